"""Independent numpy restatement of the force law (src/main.rs:234-253), vectorised over sources.

A second, separately written reading of the reference used only to cross-check the C++ oracle (two
restatements that agree bit for bit make a transcription slip in either unlikely).  Test infrastructure.
"""
import numpy as np


def pair_terms(p, src, weight, clamp=0.001, dtype=np.float32):
    """Per-source contribution to the acceleration of target p: array [n_src, 2] in `dtype`, zeros where
    the reference returns early."""
    dt = np.dtype(dtype).type
    src = np.asarray(src, dtype=dtype).reshape(-1, 2)
    p = np.asarray(p, dtype=dtype)
    with np.errstate(all="ignore"):
        diff = src - p                                           # :236
        s = np.abs(diff[:, 0]) + np.abs(diff[:, 1])              # :238
        tiny = np.finfo(dtype).tiny
        normal = np.isfinite(s) & (np.abs(s) >= tiny)            # f32::is_normal  :241
        dist = diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]  # :245
        dist = np.where(dist < dt(clamp), dt(clamp), dist)       # :247-249
        force = np.asarray(weight, dtype=np.uint32).astype(dtype)
        num = diff * force[:, None]                              # diff * force
        den = (s * dist)[:, None]                                # sum * distance
        out = num / den                                          # :252
    out = np.where(normal[:, None], out, dt(0))
    return out.astype(dtype)


def direct_accel_seq(pos, weight, targets, clamp=0.001, dtype=np.float32):
    """Sequential ascending-j accumulation in `dtype` (python loop: small inputs only)."""
    pos = np.asarray(pos, dtype=dtype).reshape(-1, 2)
    out = np.zeros((len(targets), 2), dtype)
    for k, t in enumerate(targets):
        terms = pair_terms(pos[t], pos, weight, clamp, dtype)
        # a skipped pair leaves the accumulator untouched; adding +0 to an accumulator that started at +0 is the same
        ax = dtype(0)
        ay = dtype(0)
        for j in range(pos.shape[0]):
            ax = dtype(ax + terms[j, 0])
            ay = dtype(ay + terms[j, 1])
        out[k] = (ax, ay)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Second, independent reading of the trees and the walker, in plain Python with numpy scalars (small inputs only).
# Written from the reference text (src/bvh_tree.rs, src/quad_tree.rs, src/main.rs:348-386), not from the C++ oracle.
# ---------------------------------------------------------------------------------------------------------------------
def _f(dtype):
    return np.dtype(dtype).type


def _gravity(p1, p2, acc, force, clamp, T):
    """calculate_gravity, main.rs:234-253.  acc is a 2-list of T scalars, updated in place."""
    dx, dy = T(p2[0] - p1[0]), T(p2[1] - p1[1])
    s = T(abs(dx) + abs(dy))
    tiny = np.finfo(T).tiny
    if not (np.isfinite(s) and abs(s) >= tiny):           # !sum.is_normal()
        return
    dist = T(T(dx * dx) + T(dy * dy))
    if dist < clamp:
        dist = clamp
    den = T(s * dist)
    with np.errstate(all="ignore"):
        acc[0] = T(acc[0] + T(T(dx * force) / den))
        acc[1] = T(acc[1] + T(T(dy * force) / den))


class PyBVH:
    """BVHTree::from / make_leaf / calculate_gravity (bvh_tree.rs:40-158) on a Python list of [x, y, weight, id]."""

    def __init__(self, pos, weight, leaf_size=64, dtype=np.float32):
        self.T = T = _f(dtype)
        self.leaf_size = leaf_size
        self.pts = [[T(p[0]), T(p[1]), int(w), i] for i, (p, w) in enumerate(zip(pos, weight))]
        self.root = self._from(0, len(self.pts))            # the top call is unconditional (main.rs:400)
        self._upward(self.root)

    def _fold_box(self, lo, hi):
        T = self.T
        mnx = mny = np.finfo(T).max
        mxx = mxy = T(0)
        for k in range(lo, hi):
            x, y = self.pts[k][0], self.pts[k][1]
            mnx = mnx if mnx < x else x                     # min.min(p): minps semantics
            mny = mny if mny < y else y
            mxx = mxx if mxx > x else x
            mxy = mxy if mxy > y else y
        return (mnx, mny), (T(mxx - mnx), T(mxy - mny))

    def _leaf(self, lo, hi):
        off, size = self._fold_box(lo, hi)
        return {"leaf": True, "off": off, "size": size, "lo": lo, "hi": hi}

    def _from(self, lo, hi):
        T = self.T
        n = hi - lo
        off, size = self._fold_box(lo, hi)
        sx = sy = T(0)
        for k in range(lo, hi):                              # sum.add(p.position), in slice order
            sx = T(sx + self.pts[k][0])
            sy = T(sy + self.pts[k][1])
        with np.errstate(all="ignore"):
            hx, hy = T(sx / T(n)), T(sy / T(n))
        half = n // 2
        cx = sum(1 for k in range(lo, hi) if self.pts[k][0] > hx)
        cy = sum(1 for k in range(lo, hi) if self.pts[k][1] > hy)
        hori, vert = abs(half - cx), abs(half - cy)
        axis, mean = (0, hx) if vert > hori else (1, hy)
        # partition 0.1.2: two pointers
        pts = self.pts
        split = 0
        if n > 0:
            l, r = lo, hi - 1
            while True:
                while l < hi and pts[l][axis] > mean:
                    l += 1
                while r > lo and not (pts[r][axis] > mean):
                    r -= 1
                if l >= r:
                    split = l
                    break
                pts[l], pts[r] = pts[r], pts[l]
        left = self._from(lo, split) if split - lo > self.leaf_size else self._leaf(lo, split)
        right = self._from(split, hi) if hi - split > self.leaf_size else self._leaf(split, hi)
        return {"leaf": False, "off": off, "size": size, "kids": (left, right), "cog": (T(0), T(0)), "mass": 0}

    def _cog_mass(self, node):
        T = self.T
        if node["leaf"]:
            ax = ay = T(0)
            m = 0
            for k in range(node["lo"], node["hi"]):
                ax = T(ax + self.pts[k][0])
                ay = T(ay + self.pts[k][1])
                m = (m + self.pts[k][2]) & 0xFFFFFFFF
            cnt = T(node["hi"] - node["lo"])
            with np.errstate(all="ignore"):
                return (T(ax / cnt), T(ay / cnt)), m
        return node["cog"], node["mass"]

    def _upward(self, node):
        T = self.T
        if node["leaf"]:
            return
        a, b = node["kids"]
        self._upward(a)
        self._upward(b)
        (c0, m0), (c1, m1) = self._cog_mass(a), self._cog_mass(b)
        mass = (m0 + m1) & 0xFFFFFFFF
        with np.errstate(all="ignore"):
            bx = T(T(c0[0] * T(m0)) + T(c1[0] * T(m1)))
            by = T(T(c0[1] * T(m0)) + T(c1[1] * T(m1)))
            node["cog"] = (T(bx / T(mass)), T(by / T(mass)))
        node["mass"] = mass

    def walk(self, p, theta, clamp=0.001):
        T = self.T
        acc = [T(0), T(0)]
        self._walk((T(p[0]), T(p[1])), self.root, acc, T(theta), T(np.float32(clamp)))
        return acc

    def _walk(self, p, node, acc, theta, clamp):
        T = self.T
        if node["leaf"]:
            for k in range(node["lo"], node["hi"]):
                q = self.pts[k]
                _gravity(p, (q[0], q[1]), acc, T(q[2]), clamp, T)
            return
        ox, oy = node["off"]
        w, h = node["size"]
        contains = p[1] > oy and p[0] > ox and p[0] < T(ox + w) and p[1] < T(oy + h)
        m = w if w > h else h                               # size.max(size.yx()) -> x*y
        m2 = T(m * m)
        cg = node["cog"]
        ddx, ddy = T(p[0] - cg[0]), T(p[1] - cg[1])
        d2 = T(T(ddx * ddx) + T(ddy * ddy))
        if (not contains) and m2 < T(T(d2 * theta) * theta):
            _gravity(p, cg, acc, T(node["mass"]), clamp, T)
        else:
            self._walk(p, node["kids"][0], acc, theta, clamp)
            self._walk(p, node["kids"][1], acc, theta, clamp)

    def ids(self):
        return np.array([q[3] for q in self.pts], np.uint32)


class PyQuad:
    """QuadTree::new / insert / subdivide / calculate_gravity (quad_tree.rs:55-270), points inserted in index order."""

    def __init__(self, pos, weight, root=(0.0, 0.0, 100000.0), dtype=np.float32):
        self.T = T = _f(dtype)
        self.root = self._new((T(root[0]), T(root[1])), T(root[2]))
        for i, (p, w) in enumerate(zip(pos, weight)):
            self._insert(self.root, (T(p[0]), T(p[1]), int(w), i))
        self._upward(self.root)

    def _new(self, off, h):
        return {"leaf": True, "off": off, "h": h, "pts": [], "cog": (self.T(0), self.T(0))}

    def _insert(self, node, pt):
        T = self.T
        if node["leaf"]:
            if len(node["pts"]) == 8:                       # MAX_CAPACITY
                old = node["pts"]
                node.update({"leaf": False, "kids": [None] * 4, "mass": len(old), "pts": None})
                for q in old:
                    self._insert(node, q)
                self._insert(node, pt)
            else:
                node["pts"].append(pt)
            return
        half = T(node["h"] / T(2.0))
        ox, oy = node["off"]
        xm, ym = T(ox + half), T(oy + half)
        child = (2 if pt[1] > ym else 0) + (1 if pt[0] > xm else 0)
        if node["kids"][child] is None:
            off = [(ox, oy), (T(ox + half), T(oy + T(0))), (T(ox + T(0)), T(oy + half)), (T(ox + half), T(oy + half))][child]
            node["kids"][child] = self._new(off, half)
        self._insert(node["kids"][child], pt)

    def _mass(self, node):
        if node["leaf"]:
            m = 0
            for q in node["pts"]:
                m = (m + q[2]) & 0xFFFFFFFF
            return m
        return node["mass"]

    def _upward(self, node):
        T = self.T
        if node["leaf"]:
            if node["pts"]:
                ax = ay = T(0)
                for q in node["pts"]:
                    ax = T(ax + q[0])
                    ay = T(ay + q[1])
                c = T(len(node["pts"]))
                node["cog"] = (T(ax / c), T(ay / c))
            return
        kids = [k for k in node["kids"] if k is not None]
        for k in kids:
            self._upward(k)
        mass = 0
        for k in kids:
            mass = (mass + self._mass(k)) & 0xFFFFFFFF
        bx = by = T(0)
        for k in kids:
            bx = T(bx + T(k["cog"][0] * T(self._mass(k))))
            by = T(by + T(k["cog"][1] * T(self._mass(k))))
        with np.errstate(all="ignore"):
            node["cog"] = (T(bx / T(mass)), T(by / T(mass)))
        node["mass"] = mass

    def empty(self, node=None):
        """quad_tree.rs:66-89 -> cells visited (a root whose mass is 0 already counts as one and is not entered)."""
        node = self.root if node is None else node
        if node["leaf"]:
            node["pts"] = []
            return 1
        if node["mass"] == 0:
            return 1
        node["mass"] = 0
        return 1 + sum(self.empty(k) for k in node["kids"] if k is not None)

    def prune(self, node=None):
        """quad_tree.rs:94-137 -> children dropped (empty leaves, roots without mass); roots with mass are entered."""
        node = self.root if node is None else node
        dropped = 0
        if node["leaf"]:
            return dropped
        for i, k in enumerate(node["kids"]):
            if k is None:
                continue
            if k["leaf"]:
                gone = len(k["pts"]) == 0
            elif k["mass"] == 0:
                gone = True
            else:
                gone = False
                dropped += self.prune(k)
            if gone:
                node["kids"][i] = None
                dropped += 1
        return dropped

    def reuse(self, pos, weight):
        """empty(), the points where they are now, the upward pass, prune() -> (empty()'s return, prune()'s return)."""
        T = self.T
        emptied = self.empty()
        for i, (p, w) in enumerate(zip(pos, weight)):
            self._insert(self.root, (T(p[0]), T(p[1]), int(w), i))
        self._upward(self.root)
        return emptied, self.prune()

    def cells(self, node=None, path=()):
        """The tree's shape: every cell as (path of child codes, is a leaf, ids of its points)."""
        node = self.root if node is None else node
        if node["leaf"]:
            return [(path, True, tuple(q[3] for q in node["pts"]))]
        out = [(path, False, ())]
        for c, k in enumerate(node["kids"]):
            if k is not None:
                out += self.cells(k, path + (c,))
        return out

    def walk(self, p, theta, clamp=0.001):
        T = self.T
        acc = [T(0), T(0)]
        self._walk((T(p[0]), T(p[1])), self.root, acc, T(theta), T(np.float32(clamp)))
        return acc

    def _walk(self, p, node, acc, theta, clamp):
        T = self.T
        if node["leaf"]:
            for q in node["pts"]:
                _gravity(p, (q[0], q[1]), acc, T(q[2]), clamp, T)
            return
        ox, oy = node["off"]
        h = node["h"]
        contains = p[1] > oy and p[0] > ox and p[0] < T(ox + h) and p[1] < T(oy + h)
        cg = node["cog"]
        ddx, ddy = T(p[0] - cg[0]), T(p[1] - cg[1])
        d2 = T(T(ddx * ddx) + T(ddy * ddy))
        if (not contains) and T(h * h) < T(T(d2 * theta) * theta):
            _gravity(p, cg, acc, T(node["mass"]), clamp, T)
        else:
            for k in node["kids"]:
                if k is not None:
                    self._walk(p, k, acc, theta, clamp)


def draw(pos, vel, weight, height=100_000, render_px=1250):
    """draw() of main.rs:41-72, written independently of the C++ oracle: a plain Python loop over the rows."""
    frame = np.zeros((render_px * render_px, 4), np.uint8)
    cell = height // render_px
    dt = pos.dtype.type
    for i in range(pos.shape[0]):
        x, y = pos[i]
        if not (y < dt(height) and x < dt(height) and y >= 0 and x >= 0):
            continue
        off = (int(y) // cell) * render_px + int(x) // cell
        if weight[i] > 10:
            frame[off] = (0, 255, 0, 255)
        elif frame[off, 3] != 255:
            a = (abs(vel[i, 0]) + abs(vel[i, 1])) * dt(10.0)
            b = 0 if np.isnan(a) else (255 if a >= 255 else int(a))
            v = 0x10 + min(b, 0xEF)
            frame[off, 0] = 255
            frame[off, 1] = 255 - v
            frame[off, 2] = 255 - v
            if frame[off, 3] <= 240:
                frame[off, 3] += 10
    return frame.reshape(render_px, render_px, 4)
