#!/usr/bin/env python3
"""Benchmark of the ORIGIN hot path on MI355X:  voxels/s through DCT + PCA + GLR.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            # starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path (Preprocessing -> ComputePCAThreshold ->
ComputeGreedyPCA -> ComputeTGLR dense parts) over one synthetic (Nz, N, N) cube whose
inputs (raw, var, mask) are already resident in HBM.  With N > 1 ranks the field is cut into
spatial tiles (one per GPU, PCA areas never straddle tiles) and the total work is fixed
(strong scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from origin_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA (no sparsity)
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries of this command (tools/collect_profiles.sh), newest
# first: the traffic figure of the dominant kernel is read from the first one that exists
PMC_PROFILES = ("r04_pmc_fetch_write.json", "r03_pmc_fetch_write.json", "r02_pmc_fetch_write.json")
MFMA_FLOP = 32768.0        # one v_mfma_f32_32x32x16_{f16,bf16}: 32 x 32 x 16 x 2
# kernel classes of the built-in profiler -> kernel names in the rocprofv3 summaries
PROFILE_KERNELS = {"glr_spectral": ["spectral_mfma2_kernel", "spectral3_kernel"],
                   "glr_spatial": ["spatial2_kernel", "spatial4x4_kernel"],
                   "dct_fit": ["dct_moments_kernel"], "dct_plane_sums": ["dct_part_reduce_kernel"],
                   "dct_standardize": ["dct_standardize_kernel"],
                   "pca_deflate_dot": ["deflate_dot_rows_kernel", "deflate_dot_kernel"],
                   "pca_flush": ["flush_kernel"],
                   "local_max": ["local_max3s_kernel", "local_max3v_kernel", "local_max3_kernel"]}


def executed_tflops(mfma_instructions, avg_launch_s):
    """What the matrix cores really did: MFMA instructions of one launch (the plan's own count,
    origin_glr_plan_mfma_count = rocprofv3's SQ_INSTS_MFMA) x 32768 flop over the launch time."""
    return mfma_instructions * MFMA_FLOP / avg_launch_s / 1e12


def traffic_from_profile(pmc, kernel):
    """HBM bytes per launch of `kernel` from a committed FETCH_SIZE / WRITE_SIZE summary (raw
    counter values).  FETCH_SIZE is doubled for every kernel, as MI355X_MICROARCH.md (HBM)
    prescribes for coalesced streaming reads on gfx950 -- calibrated here on the 4-byte-per-lane
    loads as well: uncorrected, dct_moments_kernel reads 6.65 GB where it must read 11.9, and
    spectral_mfma2_kernel 5.51 GB where it must read 6.63 (profiles/r02_pmc_fetch_write.json)."""
    e = pmc.get(kernel)
    if not e:
        return None
    return round((2.0 * e.get("FETCH_SIZE_GB_per_launch", 0.0) +
                  e.get("WRITE_SIZE_GB_per_launch", 0.0)) * 1e9)


def load_pmc_profile():
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PMC_PROFILES:
        try:
            return name, json.load(open(os.path.join(here, "profiles", name)))
        except (OSError, ValueError):
            continue
    return None, None


def _gen_chunk(args):
    field_args, ic, window = args
    f = _gen_chunk.cache.get(field_args)
    if f is None:
        f = _gen_chunk.cache[field_args] = synth.SyntheticField(*field_args)
    return ic, f.chunk(ic, window)


_gen_chunk.cache = {}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh interpreters of this
    script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each), relay rank 0's JSON
    line, exit non-zero as soon as any rank fails.  This parent never touches the GPU (no HIP
    call, no library load) and never replaces itself: the ranks are ordinary children."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                ORIGIN_RDV_KEY=f"bench{os.getpid()}", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else
                                      subprocess.DEVNULL))
    # rank 0's line can outgrow the pipe's buffer (64 KiB): read it while waiting, not afterwards
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        else:
            time.sleep(0.05)
    if failed is not None:
        for r in live:          # exactly the children started above
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; "
                         "the other ranks were stopped\n")
        raise SystemExit(1)
    reader.join(timeout=30)
    out = b"".join(chunks).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 did not print exactly one JSON line\n" + out[-2000:])
        raise SystemExit(1)
    print(lines[0], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=600, help="field is size x size spaxels")
    ap.add_argument("--nz", type=int, default=3681)
    ap.add_argument("--nprof", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=100,
                    help="side of the centred crop the CPU oracle is timed on (SURVEY 8d: one "
                         "100x100 area)")
    ap.add_argument("--check", choices=("off", "light", "full", "glr"), default="light",
                    help="oracle check of the arrays the timed steps produced (N=1 only): light = "
                         "one haloed GLR window + one PCA area, full = three windows + two areas "
                         "+ a DCT window, glr = the three GLR windows only "
                         "(oracle/window_check.py)")
    ap.add_argument("--e2e-size", type=int, default=300,
                    help="side of the sub-field for the PCIe-inclusive pass (host arrays in, host "
                         "arrays out through the Step seam); 0 = skip")
    ap.add_argument("--glr-precision", choices=("f16x2", "f32", "bf16"), default="f16x2")
    ap.add_argument("--no-tail-overlap", dest="tail_overlap", action="store_false",
                    help="greedy PCA, then the GLR (rounds 1-2); default: the GLR of the row bands "
                         "that do not depend on the areas still iterating starts in the shadow of "
                         "the PCA's tail (pipeline.greedy_pca_then_glr; one rank only)")
    ap.add_argument("--tail-early-budget", type=float, default=8.5e8,
                    help="voxels of GLR handed to the side stream at the tail hook at most (0: all "
                         "the bands that are ready)")
    ap.add_argument("--tail-max-active", type=int, default=2,
                    help="the tail hook fires when at most this many areas still iterate")
    ap.add_argument("--no-local-max", dest="local_max", action="store_false",
                    help="leave compute_local_max (the last dense pass of ComputeTGLR.run, reference "
                         "steps.py:796) out of the step")
    ap.set_defaults(local_max=True)
    ap.add_argument("--dense-local-max", action="store_true",
                    help="write cube_local_max / cube_local_min as two dense float32 cubes (rounds "
                         "1-3: 17 B/voxel); default: (index, value) lists of their non-zero voxels "
                         "(origin_local_max_sparse: 9 B/voxel read, one rank only)")
    ap.add_argument("--masked-border", type=int, default=0,
                    help="SURVEY 8(d)'s input variant: that many spaxels along every field edge masked "
                         "in every channel (raw 0, var inf; origin.py:262-274); the check then also "
                         "covers a DCT window on the border and the corner area (O2 == 0 spaxels)")
    ap.add_argument("--area-size", type=int, default=100,
                    help="side of the square PCA areas (development: 128 makes area rows "
                         "cache-line aligned)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    # stdout must carry exactly ONE JSON line: anything libraries print there (RCCL banners) is
    # diverted to stderr until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    Nz, N = args.nz, args.size
    field_args = (Nz, N, N, None, 25, args.nprof, args.masked_border, 1.0 / 400, 1.0 / 900,
                  args.area_size)
    # worker pool for the synthetic cube: forked BEFORE this process touches the GPU
    nworkers = max(1, min(12, (os.cpu_count() or 8) // max(1, min(world, 8)) - 1))
    pool = mp.get_context("fork").Pool(nworkers)

    # rehearsal on a 1-GPU box (ORIGIN_BENCH_SHARE_GPU=1): every rank on GPU 0, strips staged
    # through the host -- the line then says n_gpus 1 and names the transport.  Otherwise every
    # rank owns a device of its own and the strips go over RCCL, or the run fails.
    share_gpu = os.environ.get("ORIGIN_BENCH_SHARE_GPU") == "1"
    comm = None
    if world > 1:
        from origin_amd import multigpu
        comm = multigpu.init_comm(rank, world, local_rank, backend="host" if share_gpu else "rccl")

    from origin_amd import kernels, pipeline
    from origin_amd.device import Context, device_count

    if world > 1 and not share_gpu:
        ndev = device_count()
        enough = comm.group.allreduce(np.array([float(ndev >= world)]), "min")[0] == 1.0
        if not enough:
            raise SystemExit(f"bench.py --gpus {world}: rank {rank} sees {ndev} device(s); every "
                             "rank needs a GPU of its own (ORIGIN_BENCH_SHARE_GPU=1 rehearses "
                             "the tiling on one card over host-staged strips)")
    ctx = Context(local_rank if (world > 1 and not share_gpu) else 0)
    field = synth.SyntheticField(*field_args)

    if world > 1:
        # halo: the PSF's half width, plus one spaxel for the 3x3x3 local maxima of the tile
        tiling = multigpu.Tiling(field.Ny, field.Nx, world, area_size=args.area_size,
                                 halo=field.PSF.shape[-1] // 2 + (1 if args.local_max else 0))
        tile = tiling.tile(rank)
        y0, y1, x0, x1 = tile.y0, tile.y1, tile.x0, tile.x1
    else:
        tiling, tile = None, None
        y0, y1, x0, x1 = 0, N, 0, N
    ny, nx = y1 - y0, x1 - x0
    window = (y0, y1, x0, x1)

    # ---- inputs resident in HBM --------------------------------------------------
    t_gen = time.time()
    raw = ctx.empty((Nz, ny, nx), np.float32)
    var = ctx.empty((Nz, ny, nx), np.float32)
    mask = ctx.empty((Nz, ny, nx), np.uint8)
    plane = ny * nx
    jobs = [(field_args, ic, window) for ic in range(field.nchunks)]
    for ic, (r, v, m) in pool.imap_unordered(_gen_chunk, jobs):
        z0 = ic * synth.ZCHUNK
        raw.view(z0 * plane, r.shape).upload(r)
        var.view(z0 * plane, v.shape).upload(v)
        mask.view(z0 * plane, m.shape).upload(m)
    pool.close()
    pool.join()
    t_gen = time.time() - t_gen

    areamap = field.areamap[y0:y1, x0:x1]
    labels = np.unique(areamap)
    local_map = np.searchsorted(labels, areamap).astype(np.int32) + 1
    nb_local = len(labels)
    spx = pipeline.area_lists(local_map, nb_local)

    if world > 1:
        glr = multigpu.TiledGLR(ctx, comm, tiling, rank, Nz, field.PSF.astype(np.float64),
                                field.profiles, pcut=1e-8)
    else:
        plan = kernels.GLRPlan(ctx, (Nz, ny, nx), field.PSF.astype(np.float64), None,
                               field.profiles, pcut=1e-8, pmeansub=True,
                               precision=args.glr_precision)
    if comm is not None:
        comm.attach(ctx)  # RCCL communicator on this context (collective; raises on every rank
        #                   if any rank cannot create it)
        if not share_gpu and not (comm.backend == "rccl" and comm.device_p2p):
            raise SystemExit("bench.py: the strips would not go over RCCL")

    cube_std = ctx.empty((Nz, ny, nx), np.float32)
    cont_dct = ctx.empty((Nz, ny, nx), np.float32)
    cube_faint = ctx.empty((Nz, ny, nx), np.float32) if world == 1 else None  # (tiled: in glr.ext)
    # (tiled: the GLR's outputs and the local maxima stay in the tile's halo-extended arrays)
    correl = ctx.empty((Nz, ny, nx), np.float32) if world == 1 else None
    correl_min = ctx.empty((Nz, ny, nx), np.float32) if world == 1 else None
    profile = ctx.empty((Nz, ny, nx), np.uint8) if world == 1 else None
    ima_dct = ctx.empty((ny, nx), np.float32)
    ima_std = ctx.empty((ny, nx), np.float32)
    o2_buf = ctx.empty((ny, nx), np.float64)
    # work buffers of the DCT stage, allocated once (an allocation made while the GPU is busy
    # waits for it)
    coef_buf = ctx.empty((11, ny, nx), np.float64)
    zsum_buf, zcnt_buf = ctx.empty((Nz,), np.float64), ctx.empty((Nz,), np.float64)
    info = {}
    last = {}

    phase = {}
    from origin_amd.pca import GreedyPCA
    pca_driver = GreedyPCA(ctx)

    # (outputs of --local-max, allocated once: a 5 GB hipMalloc / hipFree per step is slower than
    # the kernel)
    sparse_lm = args.local_max and world == 1 and not args.dense_local_max and nx % 4 == 0
    lm_bufs = None
    if sparse_lm:
        from origin_amd import sparse
        lm_bufs = sparse.SparseBuffers(ctx, (Nz, ny, nx))
    dense_lm = args.local_max and world == 1 and not sparse_lm
    lmax_buf = ctx.empty((Nz, ny, nx), np.float32) if dense_lm else None
    lmin_buf = ctx.empty((Nz, ny, nx), np.float32) if dense_lm else None

    # tiled: the local maxima of the extended tile, as lists where that tile has a sparse form
    tiled_sparse = False
    if world > 1 and args.local_max and not args.dense_local_max:
        from origin_amd import sparse
        tiled_sparse = sparse.plan(ctx, glr.eshape)[0] > 0
    glr_key = "glr_and_local_max" if args.local_max else "glr"

    area_rows = [(int(s_.min()) // nx, int(s_.max()) // nx) if len(s_) else None for s_ in spx]
    area_boxes = [(int(s_.min()) // nx, int(s_.max()) // nx, int((s_ % nx).min()),
                   int((s_ % nx).max())) if len(s_) else None for s_ in spx]

    def one_step(with_local_max=None, overlap=None):
        do_lm = args.local_max if with_local_max is None else with_local_max
        t0 = time.perf_counter()
        ctx.aux_join()  # (coef_buf / cont_dct of the previous step: idle after its closing sync)
        coef, zsum, zcnt = kernels.dct_fit_sums(ctx, raw, var, mask, 10, False, coef=coef_buf,
                                                zsum=zsum_buf, zcnt=zcnt_buf)
        if comm is not None:
            comm.allreduce_sum_device(ctx, [zsum, zcnt])
        # cont_dct: written by the standardisation pass itself ("fused", default: var is read once),
        # or by a pass of its own after the O2 map has left -- on the main stream ("sync") or on
        # the auxiliary low-priority stream ("aux").  Measured at 3681 x 600 x 600: 54.5 / 54.3 /
        # 54.9 ms per step -- the side pass overlaps the host's threshold fit but takes its
        # 10.6 GB of traffic out of the PCA's first, HBM-bound iterations
        cont_mode = os.environ.get("ORIGIN_BENCH_CONT", "fused")
        pre = kernels.dct_standardize(ctx, raw, var, mask, coef, zsum, zcnt, cube_std=cube_std,
                                      cont_dct=cont_dct if cont_mode == "fused" else None,
                                      want_cont=cont_mode == "fused", o2=o2_buf, ima_std=ima_std)
        o2 = pre["o2"].to_host()
        # the continuum cube is not needed by anything below: it runs on the auxiliary
        # low-priority stream, under the host's threshold fit and the latency-bound kernels of the
        # greedy PCA (same step: the closing synchronisation waits for both streams)
        if cont_mode != "fused":
            kernels.dct_cont_std(ctx, var, coef, cont_dct=cont_dct, ima_dct=ima_dct,
                                 aux=cont_mode == "aux")
        t1 = time.perf_counter()
        thr = pipeline.pca_threshold(o2, local_map, nb_local, 0.01, spx=spx)
        t2 = time.perf_counter()
        overlapped = (world == 1 and plan.rows_supported() and
                      (args.tail_overlap if overlap is None else overlap))
        if overlapped:
            # the GLR of the finished part of the field starts inside the PCA's tail; t3 is taken
            # when the PCA returns (its share of the wall time includes the hook's enqueueing)
            F, mapO2, nstop, drv, out = pipeline.greedy_pca_then_glr(
                ctx, plan, cube_std, local_map, nb_local, thr["thresO2"], thr["testO2"], mask,
                correl, profile, correl_min, cube_faint, 50, 100, spx=spx, driver=pca_driver,
                o2_dev=pre["o2"], max_active=args.tail_max_active, area_rows=area_rows,
                local_max=(lm_bufs if sparse_lm else (lmax_buf, lmin_buf)) if do_lm else None,
                early_budget=args.tail_early_budget or None)
            info["glr_bands"] = {"early": out["bands"][0], "late": out["bands"][1]}
            t3 = time.perf_counter()
            ctx.sync()
            t4 = time.perf_counter()
            for k, v in (("dct_std", t1 - t0), ("threshold_fit_host", t2 - t1),
                         ("greedy_pca_and_glr_enqueued", t3 - t2), ("glr_local_max_sync", t4 - t3)):
                phase[k] = phase.get(k, 0.0) + v
            info["pca_iters"] = drv.iterations
            info["n_nuis_first"] = drv.trace[0][1] if drv.trace else 0
            info["nstop"] = nstop
            info["maxmap_max"] = float(out["maxmap"].to_host().max())
            info["area_iters_mean"] = float(np.mean([mapO2.reshape(-1)[s_].max() for s_ in spx]))
            last.update(thr=thr, mapO2=mapO2, out=out)
            return out
        # tiled: cube_faint is written straight into the interior of the GLR's halo-extended tile
        # (origin_pca_run_into) -- the tile lives there, no copy before the halo exchange; and the
        # regions of that tile which depend neither on a halo strip nor on the areas still
        # iterating start their GLR inside the PCA's tail (TiledGLR.make_tail_hook)
        hook = None
        if world > 1 and (args.tail_overlap if overlap is None else overlap):
            hook = glr.make_tail_hook(area_boxes, mask, args.tail_early_budget or None)
        if hook is not None:
            ctx.set_pca_tail_hook(hook, args.tail_max_active)
        pca_err = None
        try:
            F, mapO2, nstop, drv = pipeline.greedy_pca(
                ctx, cube_std, local_map, nb_local, thr["thresO2"], thr["testO2"], 50, 100, spx=spx,
                inplace=False, driver=pca_driver, o2_dev=pre["o2"],
                out=cube_faint if world == 1 else None,
                into=glr.faint_target() if world > 1 else None)
        except Exception as exc:   # noqa: BLE001 -- agreed on with the other ranks below
            pca_err = exc
        finally:
            if hook is not None:
                ctx.set_pca_tail_hook(None)
        if pca_err is None and hook is not None:
            pca_err = ctx.pop_tail_hook_error()
        # every rank learns whether all PCAs and hooks went through BEFORE the collective halo
        # exchange (a rank that failed would leave the others waiting in it) -- but behind the
        # regions of its own tile that need no halo: a fast rank works on those while it waits for
        # the slow ones (TiledGLR.run(before_exchange=...))
        def agree():
            ok = comm.group.allreduce(np.array([0.0 if pca_err is not None else 1.0]), "min")[0]
            if ok != 1.0:
                glr._early_done = None   # (regions the hook started belong to a step that is over)
                raise pca_err if pca_err is not None else RuntimeError(
                    "the greedy PCA or its tail hook failed on another rank")
        if pca_err is not None:
            if comm is not None:
                agree()
            raise pca_err
        t3 = time.perf_counter()
        if world > 1:
            # (no crop: correl / correl_min / profile and the local maxima stay in the tile's
            # halo-extended arrays, out["box"] is the tile inside them)
            lm_form = None if not do_lm else ("sparse" if tiled_sparse else True)
            out = glr.run(None, mask, None, None, None, local_max=lm_form, before_exchange=agree)
            info["glr_rects"] = {"ahead_of_exchange": len(glr.last_rects[0]),
                                 "behind_exchange": len(glr.last_rects[1]),
                                 "regions_in_pca_tail": glr.last_rects[2]}
        else:
            out = plan.run(cube_faint, mask=mask, correl=correl, profile=profile,
                           correl_min=correl_min, want_maps=True)
            if do_lm and sparse_lm:   # cube_local_max / cube_local_min (steps.py:796)
                out["local_max"], out["local_min"] = sparse.local_max_sparse(
                    ctx, correl, correl_min, mask, lm_bufs)
            elif do_lm:
                kernels.local_max(ctx, correl, correl_min, mask, 3, out_max=lmax_buf,
                                  out_min=lmin_buf)
        ctx.sync()
        t4 = time.perf_counter()
        for k, v in (("dct_std", t1 - t0), ("threshold_fit_host", t2 - t1),
                     ("greedy_pca", t3 - t2), (glr_key, t4 - t3)):
            phase[k] = phase.get(k, 0.0) + v
        info["pca_iters"] = drv.iterations
        info["n_nuis_first"] = drv.trace[0][1] if drv.trace else 0
        info["nstop"] = nstop
        info["maxmap_max"] = float(out["maxmap"].to_host().max())
        info["area_iters_mean"] = float(np.mean([mapO2.reshape(-1)[s].max() for s in spx]))
        last.update(thr=thr, mapO2=mapO2, out=out)
        return out

    def barrier():
        if comm is not None:
            comm.barrier()

    for _ in range(args.warmup):
        one_step()
    ctx.prof_reset()
    ctx.prof_enable(True)
    phase.clear()
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts_ = time.perf_counter()
        one_step()              # (ends with its own synchronisation)
        step_ms.append(1e3 * (time.perf_counter() - ts_))
    ctx.sync()
    barrier()
    t1 = time.perf_counter()
    ctx.prof_enable(0)
    elapsed = t1 - t0
    if comm is not None:
        elapsed = comm.max_float(elapsed)
    prof = ctx.prof_report()
    # the same number of steps WITHOUT the local maxima -- the scope of the rounds-1/2 bench line
    # (DCT + thresholds + PCA + GLR), timed the same way, so that lines of different rounds compare
    saved_phase = dict(phase)
    scope_r02 = None
    if args.local_max:
        barrier()
        ctx.sync()
        tq = time.perf_counter()
        for _ in range(args.steps):
            one_step(with_local_max=False)
        ctx.sync()
        barrier()
        el2 = time.perf_counter() - tq
        if comm is not None:
            el2 = comm.max_float(el2)
        scope_r02 = dict(step="the same steps without cube_local_max / cube_local_min (what "
                              "rounds 1-2 timed)",
                         ms_per_step=round(1e3 * el2 / max(1, args.steps), 3),
                         value=round(float(Nz) * N * N * args.steps / el2, 1))
        phase.clear()
        phase.update(saved_phase)
    # With the tail overlap the GLR's bands share the chip with the PCA's last iterations: their
    # event times are not the kernels' own.  The kernel table and the roofline block come from the
    # same number of steps in the SEQUENTIAL form (greedy PCA, then the GLR: every kernel alone on
    # the chip), timed the same way and reported next to `value`.
    sequential = None
    if world == 1 and args.tail_overlap and plan.rows_supported():
        ctx.prof_reset()
        ctx.prof_enable(True)
        ctx.sync()
        tq = time.perf_counter()
        for _ in range(args.steps):
            one_step(overlap=False)
        ctx.sync()
        el3 = time.perf_counter() - tq
        ctx.prof_enable(0)
        prof = ctx.prof_report()
        sequential = dict(step="the same steps with the GLR started behind the whole greedy PCA "
                               "(--no-tail-overlap: the form of rounds 1-3a; `roofline`, "
                               "`kernels_ms_per_step` are measured here, each kernel alone on the "
                               "chip)",
                          ms_per_step=round(1e3 * el3 / max(1, args.steps), 3),
                          value=round(float(Nz) * N * N * args.steps / el3, 1),
                          wall_ms_per_step_by_phase={
                              k_: round(1e3 * v_ / max(1, args.steps), 2) for k_, v_ in phase.items()
                              if k_ not in saved_phase or k_ in ("greedy_pca", glr_key)})
        phase.clear()
        phase.update(saved_phase)
    # per-kernel detail of the greedy PCA: one extra step OUTSIDE the timed region (an event
    # pair per PCA kernel costs ~10 us of stream time, ~5 ms per step)
    ctx.prof_reset()
    ctx.prof_enable(2)
    one_step(overlap=False)
    ctx.sync()
    ctx.prof_enable(0)
    prof_detail = ctx.prof_report()
    if sequential is not None:   # what --check looks at: the arrays of a step in the timed form
        one_step()
        ctx.sync()
    phase.clear()
    phase.update(saved_phase)
    barrier()

    total_vox = float(Nz) * N * N
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = total_vox * args.steps / elapsed

    # ---- roofline of the dominant kernel class (HIP events on the kernels' stream) ----
    local_vox = float(Nz) * ny * nx
    ntaps = sum(len(p) for p in kernels.prepare_profiles(field.profiles, 1e-8))
    algo = {
        # bytes per voxel per launch: what the kernel must move
        "dct_fit": ("hbm", 9.0 * local_vox),            # raw 4 + var 4 + mask 1
        # folded into the fit's moments pass: what is left is the partial array
        # [groups of 64 spaxels][Nz] float64, written once and read once
        "dct_plane_sums": ("hbm", 0.25 * local_vox),
        "dct_standardize": ("hbm", 17.0 * local_vox),   # + cube_std 4 + cont_dct 4
        "pca_deflate_dot": ("hbm", 4.0),                # per voxel of the launch's areas
        "pca_flush": ("hbm", 8.0 * local_vox),
        # correl 4 + correl_min 4 + mask 1 in; two dense cubes out (2 x 4) or, sparse, their
        # non-zero voxels as lists (~0.4 B/voxel, not counted)
        "local_max": ("hbm", (9.0 if sparse_lm else 17.0) * local_vox),
        # flops per launch for the compute-bound GLR stages (fp32 FMA = 2 flop)
        "glr_spatial": ("mfma", 2.0 * 25 * 25 * local_vox),
        "glr_spectral": ("mfma", 2.0 * ntaps * local_vox),
    }
    glr_precision = (glr.plan if world > 1 else plan).precision
    ext_vox = local_vox
    if world > 1:  # the GLR runs on the halo-extended tile
        ext_vox = float(Nz) * glr.eshape[1] * glr.eshape[2]
        algo["glr_spatial"] = ("mfma", 2.0 * 25 * 25 * ext_vox)
        algo["glr_spectral"] = ("mfma", 2.0 * ntaps * ext_vox)
    # dominant kernel class among those with an algorithmic work model (the PCA control
    # kernels -- select, Lanczos -- are latency chains without a meaningful roofline)
    ranked = [k for k, _ in sorted(prof.items(), key=lambda kv: -kv[1][0]) if k in algo]
    dominant = ranked[0] if ranked else None
    roofline = None
    if dominant in algo:
        bound, per_launch = algo[dominant]
        tot_ms, launches = prof[dominant]
        avg_s = tot_ms / launches / 1e3
        if dominant.startswith("pca_deflate"):
            per_launch = per_launch * local_vox  # upper bound: every area active
        extra = {}
        gplan = glr.plan if world > 1 else plan
        vox = ext_vox if world > 1 else local_vox
        # algorithmic HBM bytes of the dominant kernel's launch (SURVEY 8d asks for the byte
        # fraction next to the flop fraction for the GLR stages: C = 14 B/voxel for the spectral
        # stage -- cube_fsf 4 + mask 1 in, correl 4 + correl_min 4 + profile 1 out --, 8 for the
        # spatial one)
        algo_bytes = {"glr_spectral": 14.0 * vox, "glr_spatial": 8.0 * vox}.get(
            dominant, per_launch if bound == "hbm" else None)
        if bound == "hbm":
            ach, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            # Both GLR stages run on the f16 matrix cores (two-term split of both operands = 3
            # MFMAs per product, banded Toeplitz operand) unless the plan stays in fp32.
            # `achieved` is the ALGORITHMIC rate (2 flop per tap and voxel); `executed` counts
            # what the matrix cores really do (the plan's MFMA instruction count x 32768 flop),
            # `fp32_frac` prices the algorithmic rate against the fp32 FMA peak the same
            # problem has without them.
            on_mfma = (dominant == "glr_spectral" and glr_precision != "f32") or \
                      (dominant == "glr_spatial" and gplan.spatial_on_matrix_cores)
            ach, unit = per_launch / avg_s / 1e12, "TFLOP/s"
            peak = F16_MFMA_PEAK_TFLOPS if on_mfma else FP32_PEAK_TFLOPS
            if on_mfma:
                n_sp, n_sc = gplan.mfma_count()
                n_mfma = n_sc if dominant == "glr_spectral" else n_sp
                if n_mfma > 0:
                    ex = executed_tflops(n_mfma, avg_s)
                    extra = dict(executed=round(ex, 1), executed_frac=round(ex / peak, 4),
                                 mfma_instructions_per_launch=n_mfma,
                                 mfma_per_kvoxel=round(1024.0 * n_mfma / vox, 1))
                extra.update(fp32_frac=round(ach / FP32_PEAK_TFLOPS, 4),
                             arithmetic=glr_precision + " MFMA",
                             peak_note="dense f16/bf16 MFMA peak at 2.4 GHz; a bare MFMA chain with "
                                       "non-zero operands holds 1.98-2.11 GHz = 2.07-2.17 PFLOP/s on "
                                       "this part (power-managed clock, tools/mfma_clock_probe.hip)")
        if algo_bytes is not None:
            extra["hbm_achieved_GBs"] = round(algo_bytes / avg_s / 1e9, 1)
            extra["hbm_frac"] = round(algo_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        # HBM traffic of the dominant kernel: bench.py cannot read PMC counters itself, so it
        # takes the per-launch FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc passes
        # of this very command (profiles/, tools/summarize_rocprof.py) when the workload is the
        # default one; null otherwise
        traffic = None
        extra["traffic_measured_in_this_run"] = False
        if world == 1 and (Nz, N, args.nprof) == (3681, 600, 20):
            pname, pmc = load_pmc_profile()
            for kn in PROFILE_KERNELS.get(dominant, []) if pmc else []:
                traffic = traffic_from_profile(pmc, kn)
                if traffic is not None:
                    extra["traffic_source"] = (
                        f"profiles/{pname} (commit {pmc.get('_commit', '?')}): rocprofv3 "
                        "--pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel "
                        f"{kn}, bytes per launch, FETCH_SIZE x 2 (gfx950 correction); a counter "
                        "pass cannot run inside bench.py, so the figure is the committed "
                        "profile's, not this run's")
                    if algo_bytes:
                        extra["traffic_over_algorithmic"] = round(traffic / algo_bytes, 3)
                    break
        roofline = dict(bound=bound, kernel=dominant, achieved=round(ach, 3), peak=peak,
                        unit=unit, frac=round(ach / peak, 4), traffic=traffic,
                        avg_launch_ms=round(tot_ms / launches, 4), launches=launches, **extra)
        roofline["timed_in"] = (
            "HIP events around the kernel's launches in the sequential steps of this run (`sequential`: "
            "one launch per step, the kernel alone on the chip); in the chained steps that `value` "
            "times the stage runs as row bands beside the greedy PCA's last iterations"
            if sequential is not None else "HIP events around the kernel's launches in the timed steps")

    # ---- the greedy PCA as a whole against SURVEY 8(d)'s byte model: B = 4 (n_iter + 2) bytes per
    # voxel of an area that ran n_iter iterations (one read pass per iteration, read + write of the
    # final F = X - U C)
    if roofline is not None and "pca_total" in prof and last.get("mapO2") is not None:
        it_a = np.array([last["mapO2"].reshape(-1)[s_].max() if len(s_) else 0 for s_ in spx], float)
        n_a = np.array([len(s_) for s_ in spx], float)
        pca_bytes = float(np.sum(4.0 * (it_a + 2.0) * n_a) * Nz)
        p_ms, p_n = prof["pca_total"]
        p_s = p_ms / p_n / 1e3
        roofline["pca"] = dict(
            bound="hbm", kernel="greedy PCA, whole run (pca_total)",
            model="4 (n_iter + 2) B per voxel of an area that ran n_iter iterations",
            bytes_per_run=round(pca_bytes), ms_per_run=round(p_ms / p_n, 3),
            achieved=round(pca_bytes / p_s / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(pca_bytes / p_s / 1e9 / HBM_PEAK_GBS, 4),
            iterations_lock_step=info.get("pca_iters"),
            iterations_mean_per_area=round(float(np.average(it_a, weights=np.maximum(n_a, 1))), 2))

    # ---- whole path against the HBM roofline (the second half of BASELINE.json's metric):
    # SURVEY 8(d) algorithmic bytes per voxel, A = 17 (DCT + standardise), C = 14 (GLR),
    # B = 4 (n_iter + 2) for the voxels of an area that ran n_iter greedy-PCA iterations
    path_hbm = None
    it_mean = info.get("area_iters_mean")
    if comm is not None:  # areas are spread over the ranks: mean over all of them
        tot = comm.allreduce_sum(np.array([(it_mean or 0.0) * len(spx), float(len(spx))]))
        it_mean = float(tot[0] / max(tot[1], 1.0))
    if rank == 0 and it_mean is not None:
        lm_bpv = 0.0 if not args.local_max else (9.0 if (sparse_lm or tiled_sparse) else 17.0)
        bpv = 17.0 + 14.0 + 4.0 * (it_mean + 2.0) + lm_bpv
        gbs = bpv * Nz * N * N / (ms_per_step * 1e-3) / 1e9
        path_hbm = dict(bytes_per_voxel=round(bpv, 2), achieved=round(gbs, 1),
                        peak=HBM_PEAK_GBS * world, unit="GB/s",
                        frac=round(gbs / (HBM_PEAK_GBS * world), 4),
                        # SURVEY 8(d) asks for both peaks: the 8 TB/s of the data sheet and the
                        # 6.29 TB/s a float4 copy reaches (MI355X_MICROARCH.md)
                        frac_of_measured_copy=round(gbs / (6290.0 * world), 4),
                        note="algorithmic bytes of DCT+standardise (17 B/voxel), greedy PCA "
                             "(4 (n_iter + 2)), GLR (14)" + (f" and the 3x3x3 local maxima ({lm_bpv:g})"
                                                            if args.local_max else "") +
                             " over the step time; the GLR stages are MFMA-bound (see roofline)")

    # ---- CPU baseline: the oracle on a centred crop, all host cores, rank 0, N == 1 ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_ref
        c = min(args.cpu_crop, N)
        cy0 = (N - c) // 2
        craw, cvar, cmask = field.arrays(window=(cy0, cy0 + c, cy0, cy0 + c))
        cores = os.cpu_count() or 1
        tm = {}
        t = time.perf_counter()
        cpu_ref.run_chain(craw.astype(np.float64), cvar.astype(np.float64), cmask,
                          field.PSF.astype(np.float64), None, field.profiles,
                          np.ones((c, c), dtype=int), 1, ncpu=cores, timings=tm)
        dt = time.perf_counter() - t
        cpu_baseline = dict(value=round(Nz * c * c / dt, 1), unit="voxels/s", cores=cores,
                            kind="port",
                            sample=f"oracle.cpu_ref.run_chain on the centred {c}x{c} crop "
                                   f"({Nz}x{c}x{c} voxels, {args.nprof} profiles, 1 area), "
                                   f"GLR joblib ncpu={cores}",
                            seconds={k: round(v, 2) for k, v in tm.items()})

    # ---- oracle check of the arrays the timed steps left in HBM (rank 0, N == 1) ----------
    check = None
    if rank == 0 and world == 1 and args.check != "off":
        from oracle import window_check as wc
        t = time.perf_counter()
        full = args.check in ("full", "glr")
        psf64 = field.PSF.astype(np.float64)
        ncpu = min(32, os.cpu_count() or 1)
        out = last["out"]
        wins = wc.glr_windows(ny, nx) if full else \
            wc.glr_windows(ny, nx, out=32, halo=24, which=("interior",))
        # tolerances of the arithmetic the GLR ran in: fp32-class for f32 / f16x2,
        # screening quality for the single-bf16-MFMA form
        if glr_precision == "bf16":
            gtol = dict(tol=5e-2, tol_argmax=2e-2, tol_rms=5e-3, tol_scale_T=16.0)
            gtxt = ("GLR (bf16) |dT|<=5e-2*max(1,max_window|T|/16) = max(5e-2, 2^-8.3 of the "
                    "brightest |T| of the window) (SURVEY 8c's 5e-2 was set on a field with "
                    "T<=19.5; both operands of a product are rounded to 8 significant bits, "
                    "2*2^-9 relative to the brightest line around in the worst case; the maximum "
                    "over ~1e7 voxels came out at 2.4e-3..2.6e-3 of that line in the runs of "
                    "round 3), rms<=5e-3, argmax mismatch<=2e-2")
        else:
            gtol = dict(tol=1e-4, tol_argmax=1e-4)
            gtxt = "GLR |dT|<=1e-4, argmax mismatch<=1e-4"
        sparse_info = None
        if sparse_lm:   # the lists of the last step, made dense for the windows of the check
            sm, sn = out["local_max"], out["local_min"]
            sparse_info = dict(segments=lm_bufs.nseg, segment_capacity=lm_bufs.seg_cap,
                               fullest_segment=int(max(sm.counts().max(), sn.counts().max())),
                               local_maxima=sm.nnz, local_minima=sn.nnz,
                               list_bytes=12 * (sm.nnz + sn.nnz))
            out = dict(out, local_max=sm.dense(), local_min=sn.dense())
        elif args.local_max:   # bit-exact check of the local maxima on the same windows
            out = dict(out, local_max=lmax_buf, local_min=lmin_buf)
        glr_res = [wc.check_glr_window(cube_faint, out, mask, psf64, field.profiles, w,
                                       nthreads=ncpu, **gtol) for w in wins]
        # PCA: the area that iterated longest and (full) the one with the median count
        iters = np.array([last["mapO2"].reshape(-1)[s_].max() for s_ in spx])
        order_a = np.argsort(-iters, kind="stable")
        areas = [int(order_a[0])] + ([int(order_a[len(order_a) // 2])] if full and nb_local > 1
                                     else [])
        if not full:   # light: the median area (the longest can take minutes on the CPU)
            areas = [int(order_a[len(order_a) // 2])]
        if args.masked_border and 0 not in areas:
            areas.append(0)    # the corner area: holds fully masked spaxels (O2 == 0, lib :908-917)
        if args.check == "glr":
            areas = []
        pca_res = [wc.check_pca_area(cube_std, cube_faint, last["mapO2"], spx[a],
                                     last["thr"]["thresO2"][a], a) for a in areas]
        dct_res = []
        if args.check == "full":
            w = ("dct", ny // 2 - 8, ny // 2 + 8, nx // 3, nx // 3 + 24)
            r_ = wc.check_dct_window(raw, var, mask, cube_std, cont_dct, w)
            r_.pop("_zmean")
            dct_res.append(r_)
        if args.masked_border and args.check in ("full", "light"):
            # a window across the masked border (fully masked spaxels next to exposed ones)
            w = ("dct_border", 0, 16, nx // 2, nx // 2 + 24)
            r_ = wc.check_dct_window(raw, var, mask, cube_std, cont_dct, w)
            r_.pop("_zmean")
            dct_res.append(r_)
        check = dict(level=args.check, glr=glr_res, pca=pca_res, dct=dct_res,
                     ok=bool(all(r_["ok"] for r_ in glr_res + pca_res + dct_res)),
                     tolerances=gtxt + "; local maxima bit exact on the device's correl; PCA "
                                "rel-Frobenius<=2e-6, max-abs<=1e-4, mapO2 identical; DCT "
                                "1e-5*max(1,|x|)",
                     oracle="oracle.cpu_ref (float64) on haloed windows / whole areas of the "
                            "device arrays of the last step",
                     local_max_form=("sparse lists (origin_local_max_sparse), checked through "
                                     "origin_sparse_to_dense" if sparse_lm else "dense cubes"),
                     sparse=sparse_info,
                     seconds=round(time.perf_counter() - t, 1))

    # ---- PCIe-inclusive pass: host arrays in, host arrays out, through the Step seam ---------
    e2e = None
    if rank == 0 and world == 1 and args.e2e_size > 0:
        from origin_amd.steps import SimpleOrig, _inputs_on_device
        n_e = min(args.e2e_size, N)
        fe = synth.SyntheticField(Nz, n_e, n_e, None, 25, args.nprof, 0, 1.0 / 400, 1.0 / 900,
                                  args.area_size)
        eraw, evar, emask = fe.arrays()
        names = ("cube_std", "cube_faint", "cube_correl")
        best = None
        for _ in range(2):   # second pass: allocator and plan caches warm, as in a session
            tt = [time.perf_counter()]
            o = SimpleOrig(eraw, evar, emask, fe.PSF.astype(np.float64), fe.profiles, ctx=ctx)
            _inputs_on_device(o, ctx)      # raw, var (float32) and mask: host -> device
            ctx.sync()
            tt.append(time.perf_counter())
            o.step01_preprocessing()
            o.step02_areas.set_areamap(fe.areamap)
            o.step03_compute_PCA_threshold()
            o.step04_compute_greedy_PCA()
            o.step05_compute_TGLR()
            ctx.sync()
            tt.append(time.perf_counter())
            # float32 variant of the hand-over (what the device holds; `convert_float32` of the
            # reference's dump, steps.py:301-337): plain device -> host copies
            f32 = [o._hip_cache[n_].to_host() for n_ in names]
            tt.append(time.perf_counter())
            del f32      # (returning 16 GB of pages to the system takes a few tenths of a second:
            t_del = time.perf_counter()   # kept out of both windows)
            # what the reference's interface promises: float64 host arrays (widened natively)
            outs, per_cube = [], []
            for n_ in names:
                tq_ = time.perf_counter()
                outs.append(getattr(o, n_)._data)
                per_cube.append(round(time.perf_counter() - tq_, 3))
            outs.append(o.maxmap)
            tt.append(time.perf_counter() - (t_del - tt[-1]))   # (without the release of `f32`)
            d = np.diff(tt)
            cur = dict(h2d=d[0], steps=d[1], d2h_f32=d[2], d2h_f64=d[3], d2h_f64_per_cube=per_cube,
                       total_f64=d[0] + d[1] + d[3], total_f32=d[0] + d[1] + d[2])
            if best is None or cur["total_f64"] < best["total_f64"]:
                best = cur
            del o, outs
        vox_e = float(Nz) * n_e * n_e
        e2e = dict(value=round(vox_e / best["total_f64"], 1), unit="voxels/s",
                   seconds=round(best["total_f64"], 3),
                   value_float32_outputs=round(vox_e / best["total_f32"], 1),
                   split_seconds={k: (round(v, 3) if not isinstance(v, list) else v)
                                  for k, v in best.items()},
                   split_GBs=dict(h2d=round(9.0 * vox_e / best["h2d"] / 1e9, 1),
                                  d2h_f32=round(12.0 * vox_e / best["d2h_f32"] / 1e9, 1),
                                  d2h_f64_of_device_bytes=round(12.0 * vox_e / best["d2h_f64"] / 1e9, 1)),
                   sample=f"{Nz}x{n_e}x{n_e} sub-field: float32 host arrays in (raw, var, mask: "
                          "`h2d`), steps 1,3,4,5 of the Step seam incl. their host-side "
                          "segmentation maps, thresholds and local maxima (`steps`), host arrays "
                          "out (cube_std, cube_faint, cube_correl + maxmap): float64 as the "
                          "reference's interface holds them (`d2h_f64`: copy + native widening; "
                          "`value`) or float32 as the device holds them (`d2h_f32`; "
                          "`value_float32_outputs`); never the headline `value`")

    # ---- per-rank phase times and PCA imbalance (each rank fills its row of one all-reduce) ----
    per_rank = None
    if comm is not None:
        keys = ("dct_std", "threshold_fit_host", "greedy_pca", glr_key)
        row = np.zeros((world, len(keys) + 3))
        row[rank, :len(keys)] = [1e3 * phase.get(k_, 0.0) / max(1, args.steps) for k_ in keys]
        row[rank, len(keys):] = [info.get("pca_iters", 0), len(spx), float(ny * nx)]
        allrows = comm.allreduce_sum(row.reshape(-1)).reshape(world, -1)
        def mx_mean(v):
            v = np.asarray(v, dtype=float)
            return round(float(v.max() / max(v.mean(), 1e-300)), 3)
        per_rank = dict(phases_ms={k_: [round(v, 2) for v in allrows[:, i]]
                                   for i, k_ in enumerate(keys)},
                        imbalance=dict(note="max / mean over the ranks (1 = even)",
                                       tiling=tiling.balance(),
                                       pca_iterations=mx_mean(allrows[:, len(keys)]),
                                       **{k_: mx_mean(allrows[:, i]) for i, k_ in enumerate(keys)}),
                        pca_iterations=[int(v) for v in allrows[:, len(keys)]],
                        areas=[int(v) for v in allrows[:, len(keys) + 1]],
                        spaxels=[int(v) for v in allrows[:, len(keys) + 2]],
                        note="wall time per step on each rank between its own synchronisation "
                             "points; a rank that finishes its PCA early waits in the halo exchange "
                             "(counted in glr)")

    if rank == 0:
        line = {
            "metric": "voxels/s through DCT+PCA+GLR (ORIGIN hot path)",
            "step": "Preprocessing (DCT, standardise) -> PCA thresholds -> greedy PCA -> TGLR "
                    "(correl, correl_min, profile, maxmap, minmap" +
                    (", cube_local_max, cube_local_min" + (" as lists of their non-zero voxels)"
                                                           if sparse_lm else ")")
                     if args.local_max else ")"),
            "value": round(value, 1),
            "unit": "voxels/s",
            "n_gpus": 1 if share_gpu else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_spread": {"min": round(min(step_ms), 3),
                                   "median": round(float(np.median(step_ms)), 3),
                                   "max": round(max(step_ms), 3), "n": len(step_ms),
                                   "note": "wall time of each timed step on this rank (every step "
                                           "ends with a synchronisation)"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if glr_precision == "f32" else f"f32+{glr_precision}",
            "data": "synthetic",
            "config": {"workload": f"synthetic {Nz}x{N}x{N} cube, Dico_FWHM_2_12 "
                                   f"({args.nprof} profiles), PSF 25x25, 100x100 areas, "
                                   "dct_order 10, pfa 0.01, Noise_population 50, itermax 100",
                       "arithmetic": "DCT/PCA: f32 storage, f64 reductions and eigen-solve; GLR: "
                                     + {"f16x2": "two-term f16 split on the matrix cores, fp32 "
                                                 "accumulate (fp32-class: 22 significant bits)",
                                        "bf16": "single bf16 MFMA per product, fp32 accumulate",
                                        "f32": "fp32 FMA"}.get(glr_precision, glr_precision),
                       "masked_border": args.masked_border,
                       "glr_spectral_arithmetic": glr_precision, "tiles": world, "comm": (comm.backend + (" " + comm.note if comm.note else ""))
                       if comm is not None else None, "pca": info,
                       "gen_seconds": round(t_gen, 1)},
            "without_local_max": scope_r02,
            "sequential": sequential,
            "roofline": roofline,
            "path_hbm": path_hbm,
            "cpu_baseline": cpu_baseline,
            "check": check,
            "e2e": e2e,
            "per_rank": per_rank,
            "wall_ms_per_step_by_phase": {k: round(1e3 * v / max(1, args.steps), 2)
                                          for k, v in phase.items()},
            "kernels_ms_per_step": {k: round(v[0] / max(1, args.steps), 3)
                                    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            "kernel_launches_per_step": {k: v[1] // max(1, args.steps) for k, v in prof.items()},
            "pca_kernels_ms_detail": {k: round(v[0], 3) for k, v in
                                      sorted(prof_detail.items(), key=lambda kv: -kv[1][0])
                                      if k.startswith("pca_")},
            "pca_kernels_detail_note": "one extra step outside the timed region with an event "
                                       "pair around every PCA kernel (~10 us each)",
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if comm is not None:
        comm.close()
    # plans and the context go before the interpreter does (origin_amd.device closes contexts at
    # exit anyway; a context left to the HIP runtime's static destructors aborts the process)
    for obj in ([glr.plan] if world > 1 else [plan]):
        try:
            obj.close()
        except Exception:
            pass
    ctx.close()


if __name__ == "__main__":
    main()
