"""Area construction: ``CreateAreas.run`` (reference
muse_origin/steps.py:492-569) and the geometry behind it (lib_origin.py:92-124, :367-765).

The field is cut into areas of roughly ``minsize`` x ``minsize`` spaxels for the greedy PCA, without
cutting through continuum sources: squares -> merged when too small -> each source handed to the
square holding most of it -> convex hull of the sources of each area -> hulls grown until they
tile the exposed field -> small areas merged by compactness.  It is integer geometry on one
(Ny, Nx) map, a few dozen labels: host work, like the reference's, but it decides which GPU gets
which areas (origin_amd/multigpu.py tiles on area boundaries).

Functions keep the reference's names, arguments and return values, and the reference's behaviour
in the corner cases its tests rely on (argsort order of equal sizes, slices running off the map,
``ConvexHull`` raising on degenerate sources).  Labels are stacks ``(n_label, Ny, Nx)`` of 0/1
float64 planes as in the reference.  Two things are computed differently, with identical results:
the iterated binary dilations / erosions of ``area_growing`` (21 and 20 passes of a cross) are
taxicab distance transforms, and the closing of the hull outline is a binary dilation instead of
an FFT convolution.  Pinned by golden G10 (outputs of the reference's own functions).
"""
import numpy as np
from scipy import ndimage as ndi
from scipy.spatial import ConvexHull

_CROSS = ndi.generate_binary_structure(2, 1)


def spatial_segmentation(Nx, Ny, NbSubcube, start=None):
    """Row and column limits of NbSubcube x NbSubcube zones (lib_origin.py:92-124): rows run
    from Ny down to 0, columns from 0 up to Nx, both shifted by ``start`` = (y, x)."""
    inty = np.linspace(Ny, 0, NbSubcube + 1, dtype=int)
    intx = np.linspace(0, Nx, NbSubcube + 1, dtype=int)
    if start is not None:
        inty = inty + start[0]
        intx = intx + start[1]
    return inty, intx


def createradvar(cu, ot):
    """Variance of the distance to the centroid of ``cu`` united with each plane of ``ot``
    (lib_origin.py:367-392): the compactness the final merge minimises."""
    out = np.zeros(len(ot))
    for i, other in enumerate(ot):
        y, x = np.nonzero((cu + other) > 0)
        out[i] = np.var(np.hypot(y - y.mean(), x - x.mean()))
    return out


def _sizes(label):
    return label.reshape(len(label), -1).sum(axis=1)


def _bbox(plane, margin, shape):
    """Slices of the bounding box of the non-zero pixels of ``plane`` grown by ``margin`` and
    clipped to ``shape`` (None when the plane is empty)."""
    rows = np.flatnonzero(plane.any(axis=1))
    if not len(rows):
        return None
    cols = np.flatnonzero(plane.any(axis=0))
    return (slice(max(rows[0] - margin, 0), min(rows[-1] + margin + 1, shape[0])),
            slice(max(cols[0] - margin, 0), min(cols[-1] + margin + 1, shape[1])))


def fusion_areas(label, MinSize, MaxSize, option=None):
    """Merge every area smaller than MinSize into its best neighbour while the result stays under
    MaxSize (lib_origin.py:395-462).  Best = smallest (option None) or most compact union
    (option 'var'; the reference then compares size + *variance* with MaxSize, kept)."""
    if option not in (None, 'var'):
        raise ValueError('bad option')
    while True:
        before = label.copy()
        for n in np.argsort(_sizes(label)):
            size = label[n].sum()
            if not 0 < size < MinSize:
                continue
            box = _bbox(label[n], 1, label[n].shape)
            ring = ndi.binary_dilation(label[n][box], structure=_CROSS)
            touching = np.flatnonzero(_sizes(label[(slice(None),) + box] * ring) > 0)
            touching = touching[touching != n]
            if not len(touching):
                continue
            cost = (_sizes(label[touching]) if option is None
                    else createradvar(label[n], label[touching]))
            best = int(np.argmin(cost))
            if size + cost[best] < MaxSize:
                label[n] += label[touching[best]]
                label[touching[best]] = 0
        keep = _sizes(label) > 0
        label, before = label[keep], before[keep]
        if np.sum(before - label) == 0:
            return label


def area_segmentation_square_fusion(nexpmap, MinS, MaxS, NbSubcube, Ny, Nx):
    """Connected pieces of the exposed field inside each of the NbSubcube^2 squares, small ones
    merged (lib_origin.py:466-525).  The grid has the pitch of the full map and starts at the
    first exposed row / column, so its last squares run off the map and are clipped."""
    rows = np.flatnonzero(nexpmap.sum(axis=1) > 0)
    cols = np.flatnonzero(nexpmap.sum(axis=0) > 0)
    inty, intx = spatial_segmentation(Nx, Ny, NbSubcube, start=(rows[0], cols[0]))
    planes = []
    for numy in range(NbSubcube):
        for numx in range(NbSubcube):
            box = (slice(inty[numy + 1], inty[numy]), slice(intx[numx], intx[numx + 1]))
            tile = nexpmap[box]
            if np.mean(tile) != 0:
                pieces, npieces = ndi.label(tile)
                for p in range(1, npieces + 1):
                    plane = np.zeros((Ny, Nx))
                    plane[box] = pieces == p
                    planes.append(plane)
    return fusion_areas(np.array(planes), MinS, MaxS)


def area_segmentation_sources_fusion(labsrc, label, pfa, Ny, Nx):
    """Give every continuum source (label of ``labsrc``) whole to the area overlapping it most and
    take its pixels away from the others (lib_origin.py:528-585).  Returns the areas and the
    map of source pixels.  ``label`` is modified in place, like the reference's."""
    nsrc = int(labsrc.max())
    covered = np.zeros((Ny, Nx))
    # disjoint areas (what area_segmentation_square_fusion produces) stay disjoint, and then only
    # the pixels of the source change hands: work on those instead of on whole planes
    disjoint = bool(len(label)) and label.sum(axis=0).max() <= 1
    for s in range(1, nsrc + 1):
        src = (labsrc == s).astype(float)
        covered += src
        if not len(label):
            continue
        if disjoint:
            ys, xs = np.nonzero(src)
            at_src = label[:, ys, xs]
            owner = int(np.argmax(at_src.sum(axis=1)))
            label[:, ys, xs] = 0
            label[owner, ys, xs] = 1
        else:
            owner = int(np.argmax(_sizes(label * src)))
            label[owner] = (label[owner] + src) > 0
            others = np.arange(len(label)) != owner
            label[others] *= 1 - label[owner]
    return label, covered


def Convexline(points, snx, sny):
    """Filled convex hull of integer points (row, column) on a (max row + 1, max column + 1)
    grid (lib_origin.py:630-690): hull edges rasterised along their longer axis (end point
    excluded, ordinate truncated), outline thickened by a cross, rows filled between their first
    and last outline pixel."""
    hull = ConvexHull(points)
    ends_a = hull.points[hull.simplices[:, 1]]
    ends_b = hull.points[hull.simplices[:, 0]]
    sny, snx = points[:, 0].max() + 1, points[:, 1].max() + 1
    outline = np.zeros((sny, snx))
    for (ya, xa), (yb, xb) in zip(ends_a, ends_b):
        steep = abs(yb - ya) > abs(xb - xa)
        (u0, v0), (u1, v1) = ((ya, xa), (yb, xb)) if steep else ((xa, ya), (xb, yb))
        if u0 > u1:
            u0, v0, u1, v1 = u1, v1, u0, v0
        u = np.arange(u0, u1, dtype=int)
        v = np.array(v0 + (u - u0) * (v1 - v0) / len(u), dtype=int) if len(u) else u
        if steep:
            outline[u, v] = 1
        else:
            outline[v, u] = 1
    closed = ndi.binary_dilation(outline, structure=_CROSS)
    filled = closed.copy()
    for row_in, row_out in zip(closed, filled):
        on = np.flatnonzero(row_in)
        row_out[on[0]:on[-1]] = True
    return filled


def area_segmentation_convex_fusion(label, src):
    """Filled convex hull of the source pixels of each area, clipped to the area
    (lib_origin.py:588-627).  Areas without a source pixel are dropped."""
    hulls = []
    for plane in label:
        pts = np.argwhere(src * plane > 0)
        if not len(pts):
            continue
        y0, x0 = pts.min(axis=0)
        pts = pts - (y0, x0)
        sny, snx = pts.max(axis=0) + 1
        full = np.zeros(plane.shape)
        full[y0:y0 + sny, x0:x0 + snx] = Convexline(pts, snx, sny)
        hulls.append(full * plane)
    return np.array(hulls)


def _dilate_taxicab(plane, radius):
    """``binary_dilation(plane, iterations=radius)`` with the default cross: every pixel within
    taxicab distance ``radius`` of the set."""
    if not plane.any():
        return np.zeros(plane.shape, bool)
    return ndi.distance_transform_cdt(plane == 0, metric='taxicab') <= radius


def _erode_taxicab(plane, radius):
    """``binary_erosion(plane, border_value=1, iterations=radius)``: pixels farther than
    ``radius`` (taxicab) from every unset pixel of the map; beyond the map counts as set."""
    if plane.all():
        return np.ones(plane.shape, bool)
    return ndi.distance_transform_cdt(plane != 0, metric='taxicab') > radius


def area_growing(label, mask):
    """Grow the hulls, smallest first, until they tile ``mask`` (lib_origin.py:693-731): each
    pass closes an area over 20 pixels and dilates it by one more, restricted to exposed pixels no
    other area holds; stops when the field is covered or nothing changes."""
    order = np.argsort(_sizes(label))
    niter = 20
    grown = label.copy()
    shape = grown.shape[1:]
    held = grown.sum(axis=0)  # how many areas hold each pixel (kept up to date below)
    tight = [_bbox(plane, 0, shape) for plane in grown]  # bounding box of each area
    target = np.sum(mask)
    while True:
        total = held.sum()
        for n in order:
            if tight[n] is None:
                continue
            # the closed + dilated area lies within niter + 1 pixels of the area: work in that
            # window, one more pixel wide so that the erosion sees the unset ring around it
            # (beyond the map counts as set, which is also what the transform assumes)
            (r0, r1), (c0, c1) = [(sl.start, sl.stop) for sl in tight[n]]
            box = (slice(max(r0 - niter - 2, 0), min(r1 + niter + 2, shape[0])),
                   slice(max(c0 - niter - 2, 0), min(c1 + niter + 2, shape[1])))
            old = grown[n][box]
            free = (1 - (held[box] - old > 0)) * mask[box]
            closed = _erode_taxicab(_dilate_taxicab(old != 0, niter + 1), niter)
            new = closed * free
            held[box] += new - old
            grown[n][box] = new
            inner = _bbox(new, 0, new.shape)
            tight[n] = None if inner is None else tuple(
                slice(b.start + i.start, b.start + i.stop) for b, i in zip(box, inner))
        if held.sum() == target or held.sum() == total:
            return grown


def area_segmentation_final(label, MinS, MaxS):
    """Merge the areas that are still too small by compactness and number them
    (lib_origin.py:734-765)."""
    label = fusion_areas(label, MinS, MaxS, option='var')
    areamap = np.zeros(label.shape[1:])
    for i, plane in enumerate(label):
        areamap[plane > 0] = i + 1
    return areamap


def create_areamap(mask, segmap_merged, pfa=0.2, minsize=100, maxsize=None):
    """Body of ``CreateAreas.run`` (steps.py:530-563) on plain arrays: cube mask (Nz, Ny, Nx) or
    exposure map (Ny, Nx), merged segmentation map -> (areamap int (Ny, Nx), nbAreas)."""
    mask = np.asarray(mask)
    nexpmap = ((~mask.astype(bool)).sum(axis=0) > 0).astype(int) if mask.ndim == 3 \
        else (mask > 0).astype(int)
    Ny, Nx = nexpmap.shape
    NbSubcube = np.maximum(1, int(np.sqrt(np.sum(nexpmap) / (minsize ** 2))))
    if NbSubcube > 1:
        if maxsize is None:
            maxsize = minsize * 2
        MinSize, MaxSize = minsize ** 2, maxsize ** 2
        squares = area_segmentation_square_fusion(nexpmap, MinSize, MaxSize, NbSubcube, Ny, Nx)
        with_src, src = area_segmentation_sources_fusion(np.asarray(segmap_merged), squares, pfa,
                                                         Ny, Nx)
        hulls = area_segmentation_convex_fusion(with_src, src)
        grown = area_growing(hulls, nexpmap)
        areamap = area_segmentation_final(grown, MinSize, MaxSize)
    else:
        areamap = nexpmap
    areamap = areamap.astype(int)
    labels = np.unique(areamap)
    return areamap, len(labels) - (1 if 0 in labels else 0)
