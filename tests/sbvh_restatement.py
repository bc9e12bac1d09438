"""A second, independent restatement of the reference's SBVH builder (src/sbvh.cpp:3-449, with the pieces of src/bvh.cpp, src/bvhnode.hpp,
src/rtutil.hpp and include/math it uses), in plain Python over numpy.float32 scalars: every arithmetic operation rounds to fp32 where
the C++ does.  Test infrastructure only (tests/test_host.py compares host/bvh.cpp's Mode::SBVH with it on small scenes); far too slow
for anything else.  The reference's sbvh.cpp cannot be built here (progressview.hpp -> nanogui), so this is a cross-check between two
restatements, not a pin against the reference's object code.

Follows, line by line: the constructor (:3-50), build (:105-157), sahSplit (:159-223) with BVH::sortReferences (src/bvh.cpp:258-272),
partitionObject (:226-238), binSplit (:243-324), partitionSpatial (:328-407), splitReference (:410-449), createLeaf (:90-102),
convertTree (:52-73).  Parameters: src/sbvh.hpp:36-43, :70 (MaxLeafElems 8, MinLeafElems 1, MaxDepth 64, MaxSpatialDepth 48, 128 bins,
splitAlpha 1e-5), costTri = 1 (src/bvh.hpp:72-74)."""
import numpy as np

f32 = np.float32
FLT_MAX = f32(3.402823466e+38)
MAX_LEAF, MIN_LEAF, MAX_DEPTH, MAX_SPATIAL_DEPTH, BINS = 8, 1, 64, 48, 128
INT_MIN = -(1 << 31)


def _min(a, b):            # std::min(a, b) = (b < a) ? b : a
    return b if b < a else a


def _max(a, b):            # std::max(a, b) = (a < b) ? b : a
    return b if a < b else a


class Box:
    __slots__ = ("mn", "mx")

    def __init__(self, mn=None, mx=None):
        self.mn = [FLT_MAX] * 3 if mn is None else list(mn)
        self.mx = [-FLT_MAX] * 3 if mx is None else list(mx)

    def copy(self):
        return Box(self.mn, self.mx)

    def expand_box(self, b):
        for k in range(3):
            self.mn[k] = _min(self.mn[k], b.mn[k])          # vmin / vmax are std::min / std::max per component
            self.mx[k] = _max(self.mx[k], b.mx[k])

    def expand_point(self, p):
        for k in range(3):
            self.mn[k] = _min(self.mn[k], p[k])
            self.mx[k] = _max(self.mx[k], p[k])

    def intersect(self, b):
        for k in range(3):
            self.mn[k] = _max(self.mn[k], b.mn[k])
            self.mx[k] = _min(self.mx[k], b.mx[k])

    def area(self):        # src/rtutil.hpp:27-30
        d = [f32(self.mx[k] - self.mn[k]) for k in range(3)]
        return f32(f32(2) * f32(f32(f32(d[0] * d[1]) + f32(d[0] * d[2])) + f32(d[1] * d[2])))


class Ref:
    __slots__ = ("ind", "box")

    def __init__(self, ind, box):
        self.ind, self.box = ind, box


def _to_int(x):            # (int)float: truncation; out of range / NaN -> INT_MIN (cvttss2si)
    if not np.isfinite(x) or abs(float(x)) >= 2147483648.0:
        return INT_MIN
    return int(x)


class SBVH:
    def __init__(self, verts):
        """verts: (ntris, 3, 3) float32 positions."""
        self.V = [[[f32(c) for c in v] for v in t] for t in verts]
        n = len(self.V)
        self.refs = []
        root = Box()
        for i, t in enumerate(self.V):
            b = Box([_min(t[0][k], _min(t[1][k], t[2][k])) for k in range(3)], [_max(t[0][k], _max(t[1][k], t[2][k])) for k in range(3)])
            self.refs.append(Ref(i, b))
            root.expand_box(b)
        self.right_boxes = [None] * (max(n, BINS) - 1)
        self.min_overlap = f32(root.area() * f32(1e-5))
        self.indices = []
        self.depth = self.splits = self.duplicates = self.spatial = 0
        with np.errstate(all="ignore"):
            tree = self.build(n, root, 0)
        self.indices.reverse()
        self.nodes = []                                    # (box, parent, iStart | rightChild, nPrims)
        self.convert(tree, -1)

    # ---- :105-157
    def build(self, nrefs, box, depth):
        self.depth = max(self.depth, depth)
        if nrefs <= MIN_LEAF or depth >= MAX_DEPTH:
            return self.leaf(nrefs, box)
        parent_area = box.area()
        node_sah = f32(f32(parent_area * f32(2)) * f32(1))
        obj = self.sah_split(nrefs, node_sah)
        spatial = dict(cost=FLT_MAX, dim=-1, pos=f32(0))
        if depth < MAX_SPATIAL_DEPTH:
            ov = obj["lb"].copy()
            ov.intersect(obj["rb"])
            if ov.area() >= self.min_overlap:
                spatial = self.bin_split(nrefs, box, node_sah)
        parent_cost = f32(f32(parent_area * f32(nrefs)) * f32(1))
        min_cost = _min(obj["cost"], _min(spatial["cost"], parent_cost))
        if min_cost == parent_cost and nrefs <= MAX_LEAF:
            return self.leaf(nrefs, box)
        ln = rn = 0
        lb = rb = None
        if min_cost == spatial["cost"]:
            ln, lb, rn, rb = self.partition_spatial(nrefs, spatial)
            self.spatial += 1 if (ln and rn) else 0
        if not ln or not rn:
            ln, lb, rn, rb = self.partition_object(nrefs, obj)
        self.splits += 1
        self.duplicates += ln + rn - nrefs
        right = self.build(rn, rb, depth + 1)              # right first: duplicates go to the end of the reference stack
        left = self.build(ln, lb, depth + 1)
        return ("inner", box, left, right)

    # ---- :90-102
    def leaf(self, nrefs, box):
        for _ in range(nrefs):
            self.indices.append(self.refs.pop().ind)
        return ("leaf", box, len(self.indices) - nrefs, len(self.indices))

    # ---- src/bvh.cpp:258-272
    def sort_refs(self, start, end, dim):
        seg = self.refs[start:end + 1]
        seg.sort(key=lambda r: (f32(r.box.mn[dim] + r.box.mx[dim]), r.ind))
        self.refs[start:end + 1] = seg

    # ---- :159-223
    def sah_split(self, nrefs, node_sah):
        best_tie = FLT_MAX
        info = dict(cost=FLT_MAX, i=-1, dim=-1, lb=Box(), rb=Box())
        start = len(self.refs) - nrefs
        end = len(self.refs) - 1
        for dim in range(3):
            self.sort_refs(start, end, dim)
            rbounds = Box()
            for i in range(nrefs - 1, 0, -1):
                rbounds.expand_box(self.refs[start + i].box)
                self.right_boxes[i - 1] = rbounds.copy()
            lbox = Box()
            left_count = 0
            for i in range(1, nrefs):
                lbox.expand_box(self.refs[start + i - 1].box)
                left_count += 1
                rbox = self.right_boxes[i - 1]
                area_r, area_l = rbox.area(), lbox.area()
                cleft = f32(f32(area_l * f32(left_count)) * f32(1))
                cright = f32(f32(area_r * f32(nrefs - left_count)) * f32(1))
                cost = f32(f32(node_sah + cleft) + cright)
                tie = f32(float(i) ** 2 + float(nrefs - i) ** 2)        # pow(float, int) is double arithmetic, stored to F32
                if cost < info["cost"] or (cost == info["cost"] and tie < best_tie):
                    info = dict(cost=cost, i=i, dim=dim, lb=lbox.copy(), rb=rbox.copy())
                    best_tie = tie
        return info

    # ---- :226-238
    def partition_object(self, nrefs, info):
        start = len(self.refs) - nrefs
        self.sort_refs(start, len(self.refs) - 1, info["dim"])
        return info["i"], info["lb"], nrefs - info["i"], info["rb"]

    # ---- :410-449
    def split_reference(self, ref, dim, coord):
        left, right = Box(), Box()
        t = self.V[ref.ind]
        offsets = (2, 0, 1)
        for i in range(3):
            p1, p2 = t[offsets[i]], t[i]
            v0p, v1p = p1[dim], p2[dim]
            if v0p <= coord:
                left.expand_point(p1)
            if v0p >= coord:
                right.expand_point(p1)
            if (v0p < coord and v1p > coord) or (v0p > coord and v1p < coord):
                w = _max(f32(0), _min(f32(1), f32(f32(coord - v0p) / f32(v1p - v0p))))
                omw = f32(f32(1) - w)
                pt = [f32(f32(p1[k] * omw) + f32(p2[k] * w)) for k in range(3)]       # lerp: a * (1 - t) + b * t
                left.expand_point(pt)
                right.expand_point(pt)
        left.mx[dim] = coord
        right.mn[dim] = coord
        left.intersect(ref.box)
        right.intersect(ref.box)
        return Ref(ref.ind, left), Ref(ref.ind, right)

    # ---- :243-324
    def bin_split(self, nrefs, box, node_sah):
        origin = box.mn
        bin_size = [f32(f32(box.mx[k] - origin[k]) * f32(f32(1) / f32(BINS))) for k in range(3)]
        inv = [f32(f32(1) / bin_size[k]) for k in range(3)]
        bins = [[[Box(), 0, 0] for _ in range(BINS)] for _ in range(3)]
        for ref in self.refs[len(self.refs) - nrefs:]:
            first = [max(0, min(BINS - 1, _to_int(f32(f32(ref.box.mn[k] - origin[k]) * inv[k])))) for k in range(3)]
            last = [max(first[k], min(BINS - 1, _to_int(f32(f32(ref.box.mx[k] - origin[k]) * inv[k])))) for k in range(3)]
            for dim in range(3):
                cur = ref
                for i in range(first[dim], last[dim]):
                    coord = f32(origin[dim] + f32(bin_size[dim] * f32(i + 1)))
                    l, r = self.split_reference(cur, dim, coord)
                    bins[dim][i][0].expand_box(l.box)
                    cur = r
                bins[dim][last[dim]][0].expand_box(cur.box)
                bins[dim][first[dim]][1] += 1
                bins[dim][last[dim]][2] += 1
        split = dict(cost=FLT_MAX, dim=-1, pos=f32(0))
        for dim in range(3):
            rbounds = Box()
            for i in range(BINS - 1, 0, -1):
                rbounds.expand_box(bins[dim][i][0])
                self.right_boxes[i - 1] = rbounds.copy()
            lbounds = Box()
            left_num, right_num = 0, nrefs
            for i in range(1, BINS):
                lbounds.expand_box(bins[dim][i - 1][0])
                left_num += bins[dim][i - 1][1]
                right_num -= bins[dim][i - 1][2]
                la, ra = lbounds.area(), self.right_boxes[i - 1].area()
                sah = f32(f32(node_sah + f32(f32(la * f32(left_num)) * f32(1))) + f32(f32(ra * f32(right_num)) * f32(1)))
                if sah < split["cost"]:
                    split = dict(cost=sah, dim=dim, pos=f32(origin[dim] + f32(bin_size[dim] * f32(i))))
        return split

    # ---- :328-407
    def partition_spatial(self, nrefs, split):
        R = self.refs
        dim, pos = split["dim"], split["pos"]
        left_start = len(R) - nrefs
        left_end = left_start
        right_start = len(R)
        lbox, rbox = Box(), Box()
        i = left_end
        while i < right_start:
            if R[i].box.mx[dim] <= pos:
                lbox.expand_box(R[i].box)
                R[i], R[left_end] = R[left_end], R[i]
                left_end += 1
            elif R[i].box.mn[dim] >= pos:
                rbox.expand_box(R[i].box)
                right_start -= 1
                R[i], R[right_start] = R[right_start], R[i]
                i -= 1
            i += 1
        while left_end < right_start:
            lref, rref = self.split_reference(R[left_end], dim, pos)
            lub, rub, ldb, rdb = lbox.copy(), rbox.copy(), lbox.copy(), rbox.copy()
            lub.expand_box(R[left_end].box)
            rub.expand_box(R[left_end].box)
            ldb.expand_box(lref.box)
            rdb.expand_box(rref.box)
            lac = f32(f32(1) * f32(left_end - left_start))
            rac = f32(f32(1) * f32(len(R) - right_start))
            lbc = f32(f32(1) * f32(left_end - left_start + 1))
            rbc = f32(f32(1) * f32(len(R) - right_start + 1))
            unsplit_left = f32(f32(lub.area() * lbc) + f32(rbox.area() * rac))
            unsplit_right = f32(f32(lbox.area() * lac) + f32(rub.area() * rbc))
            duplicate = f32(f32(ldb.area() * lbc) + f32(rdb.area() * rbc))
            m = _min(unsplit_left, _min(unsplit_right, duplicate))
            if m == unsplit_left:
                lbox = lub
                left_end += 1
            elif m == unsplit_right:
                rbox = rub
                right_start -= 1
                R[left_end], R[right_start] = R[right_start], R[left_end]
            else:
                lbox, rbox = ldb, rdb
                R[left_end] = lref
                left_end += 1
                R.append(rref)
        return left_end - left_start, lbox, len(R) - right_start, rbox

    # ---- :52-73
    def convert(self, node, parent):
        ind = len(self.nodes)
        self.nodes.append(None)
        if node[0] == "leaf":
            _, box, lo, hi = node
            self.nodes[ind] = (box, parent, len(self.indices) - hi, hi - lo)
        else:
            _, box, left, right = node
            self.convert(left, ind)
            rc = len(self.nodes)
            self.convert(right, ind)
            self.nodes[ind] = (box, parent, rc, 0)
