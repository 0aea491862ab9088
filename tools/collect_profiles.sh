#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02
# Writes gpurun_out/<tag>_*: bench line, rocprofv3 kernel stats, FETCH/WRITE and MFMA counter
# passes (each --pmc pass on its own, with --kernel-trace only), PCA timeline.
set -o pipefail
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (--no-tail-overlap: the sequential form of the step -- one launch per kernel and step, every kernel
# alone on the chip; the default form splits the GLR into row bands beside the PCA's tail)
B="bench.py --steps 5 --warmup 2 --check off --no-cpu-baseline --e2e-size 0 --no-tail-overlap"
python3 $B > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || exit 1
rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_kt -- python3 $B > $out/${tag}_kt.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/${tag}_fetch -- python3 $B > $out/${tag}_fetch.log 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/${tag}_write -- python3 $B > $out/${tag}_write.log 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -f csv -d $out/${tag}_mfma -- python3 $B > $out/${tag}_mfma.log 2>&1 || exit 5
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $out/${tag}_lds -- python3 $B > $out/${tag}_lds.log 2>&1 || exit 6
python3 tools/summarize_rocprof.py $out/${tag}_kt $out/${tag}_fetch $out/${tag}_write $out/${tag} > $out/${tag}_summary.txt 2>&1
python3 tools/pmc_summary.py $out/${tag}_glr_pmc.json $out/${tag}_mfma $out/${tag}_lds >> $out/${tag}_summary.txt 2>&1
kt=$(ls $out/${tag}_kt/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$kt" ] && python3 tools/timeline_gaps.py "$kt" > $out/${tag}_pca_timeline.txt 2>&1
# the raw traces are large: keep the summaries only
rm -rf $out/${tag}_kt $out/${tag}_fetch $out/${tag}_write $out/${tag}_mfma $out/${tag}_lds
head -30 $out/${tag}_summary.txt
