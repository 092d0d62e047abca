"""Fourier series container + its device-resident twin and cached PTR rules.

Mirrors FourierSeries / FourierWorkspace as the reference uses them (FourierSeriesEvaluators v1
semantics, ref docs/src/examples.md:26-42, src/fourier.jl:56-86): all arithmetic happens in
libabzhip.so on the GPU.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib as L


class FourierSeries:
    """s(x) = sum_i c[i] exp(2 pi i sum_j (i_j + o_j) x_j / t_j), i_j 1-based like Julia.

    `c`: array of shape (M_1..M_d) (scalar series) or (M_1..M_d, n, n) (matrix valued);
    `period`: scalar or d-tuple; `offset`: FourierSeriesEvaluators' offset (index shift), or give
    `first` = the integer frequency of c[0] along each dim (an OffsetArray's first axis value).
    ref: test/dos.jl:114 (`period=, offset=`), test/utils.jl:3-9, aps_example/aps_example.jl:15-27.
    """

    def __init__(self, c, period=1.0, offset=0, *, first=None, ndim=None):
        c = np.asarray(c)
        if ndim is None:
            ndim = c.ndim - 2 if (c.ndim >= 3 and c.shape[-1] == c.shape[-2]) else c.ndim
        self.d = int(ndim)
        if not 1 <= self.d <= 3:
            raise ValueError("FourierSeries: 1 <= ndim <= 3 supported")
        self.scalar = c.ndim == self.d
        if self.scalar:
            c = c.reshape(c.shape + (1, 1))
        if c.ndim != self.d + 2 or c.shape[-1] != c.shape[-2]:
            raise ValueError("coefficient array must be (M_1..M_d) or (M_1..M_d, n, n)")
        self.c = np.array(c, dtype=np.complex128)
        self.n = int(c.shape[-1])
        self.t = tuple(float(p) for p in (period if np.ndim(period) else (period,) * self.d))
        if first is not None:
            self.first = tuple(int(v) for v in (first if np.ndim(first) else (first,) * self.d))
        else:
            off = offset if np.ndim(offset) else (offset,) * self.d
            self.first = tuple(int(o) + 1 for o in off)
        self._dev = {}

    # fields named like the reference (test/dos.jl:122,129)
    @property
    def o(self):
        return tuple(f - 1 for f in self.first)

    @property
    def dims(self):
        return self.c.shape[: self.d]

    def invalidate(self):
        """Call after mutating `c` in place (ref: DOSCache.isfresh, test/dos.jl:123-124): every device copy
        gets the new coefficients and its cached rules are re-evaluated before their next use.  Handles held
        elsewhere (a solver's cacheval, a second DOSCache on the same series) stay valid."""
        self._dev = {k: dev for k, dev in self._dev.items() if dev._h is not None}
        for dev in self._dev.values():
            dev.update()

    def device(self, ctx=None):
        ctx = ctx or L.Context.default()
        dev = self._dev.get(id(ctx))
        if dev is None:
            dev = DeviceSeries(self, ctx)
            self._dev[id(ctx)] = dev
        return dev

    def __call__(self, x):
        """Evaluate at one point (the fallback evaluator, ref src/fourier.jl:120-122)."""
        v = self.device().eval_nodes(np.atleast_2d(np.asarray(x, dtype=np.float64)))[0]
        return v


def julia_coefficient_order(c, d):
    """(M_1..M_d, n, n) numpy array -> flat complex array in the reference's memory order
    (block column-major, i_1 fastest ... i_d slowest)."""
    axes = tuple(range(d - 1, -1, -1)) + (d + 1, d)
    return np.ascontiguousarray(np.transpose(c, axes)).reshape(-1)


class DeviceSeries:
    """abz_series handle + cache of device-resident rules keyed by (npt, symmetry set, want)."""

    def __init__(self, s: FourierSeries, ctx):
        self.s = s
        self.ctx = ctx
        flat = julia_coefficient_order(s.c, s.d)
        buf = np.ascontiguousarray(flat.view(np.float64))
        _, pbuf = L.f64(buf)
        dims, pdims = L.i32(np.array(s.dims, dtype=np.int32))
        first, pfirst = L.i32(np.array(s.first, dtype=np.int32))
        per, pper = L.f64(np.array(s.t))
        h = C.c_void_p()
        L.check(L.lib().abz_series_create(ctx.h, pbuf, s.d, pdims, pfirst, pper, s.n, C.byref(h)))
        self.h = h
        self.rules = {}
        self.kshard = None     # (rank, world): rules hold this rank's share of the nodes (dist.kshard)
        self.allreduce = None  # callable summing a float64 array over the ranks of the shard group
        self.rule_bytes = 0
        self.max_rule_bytes = 96 << 30  # keep rules resident in the 288 GB of HBM, LRU beyond this
        self.generation = 0  # bumped by update(): rules evaluated from older coefficients refill before use
        # the finalizer holds the raw handle only: passing `self.rules` kept every dropped series alive for ever
        # (registry -> rules -> rule.dev -> this object); the library reference-counts series <- rule, so the rules'
        # own finalizers may run before or after this one
        self._fin = weakref.finalize(self, DeviceSeries._destroy, h)

    @staticmethod
    def _destroy(h):
        try:
            if h:
                L.lib().abz_series_destroy(h)
        except Exception:
            pass

    def close(self):
        for r in list(self.rules.values()):
            r.close()
        self.rules.clear()
        self.rule_bytes = 0
        self._fin()
        self._h = None

    @property
    def h(self):
        if self._h is None:
            raise L.AbzError("DeviceSeries was closed")
        return self._h

    @h.setter
    def h(self, v):
        self._h = v

    def update(self, c=None):
        """Upload new coefficients of the same shape.  Every cached rule is stale from here on and is
        re-evaluated in place (abz_rule_rebuild, which also refreshes its Hermitian flag) before its next
        use -- the reference rebuilds its rule from the current series on every solve.  New coefficients
        `c` replace the host series' array and go to EVERY live device copy of the series (one per context
        it was used on): a copy left behind would integrate the old coefficients."""
        if c is not None:
            c = np.asarray(c, dtype=np.complex128)
            if c.shape != self.s.c.shape:
                c = c.reshape(self.s.c.shape)
            self.s.c = np.array(c)
            for dev in list(self.s._dev.values()):
                if dev is not self and dev._h is not None:
                    dev._upload()
        self._upload()

    def _upload(self):
        buf = np.ascontiguousarray(julia_coefficient_order(self.s.c, self.s.d).view(np.float64))
        L.check(L.lib().abz_series_update(self.h, buf.ctypes.data_as(L.c_f64p)))
        self.generation += 1
        if any(r.want & L.WANT_H_COMPACT for r in self.rules.values()) and not self.hermitian():
            self.drop_rules()  # upper-triangle rules cannot hold the values of a series that stopped being Hermitian

    # ---- arbitrary nodes (BatchIntegrand body / fallback evaluator)
    def eval_nodes(self, k, want=L.WANT_H):
        k = np.ascontiguousarray(np.asarray(k, dtype=np.float64).reshape(-1, self.s.d))
        nk = len(k)
        n = self.s.n
        H = np.empty((nk, n * n, 2)) if want & L.WANT_H else None
        E = np.empty((nk, n)) if want & L.WANT_EIG else None
        pk = k.ctypes.data_as(L.c_f64p)
        pH = H.ctypes.data_as(L.c_f64p) if H is not None else None
        pE = E.ctypes.data_as(L.c_f64p) if E is not None else None
        # (row-major matrices straight from the device: transposing the reference's column-major blocks here cost more than
        # evaluating them -- 4 096 matrices of 32 x 32: tens of ms)
        L.check(L.lib().abz_eval_nodes(self.h, pk, nk, want | (L.WANT_H_ROW_MAJOR if H is not None else 0), pH, pE))
        out = []
        if H is not None:
            Hc = H.view(np.complex128).reshape(nk, n, n)
            out.append(Hc[:, 0, 0] if self.s.scalar else Hc)
        if E is not None:
            out.append(E)
        return out[0] if len(out) == 1 else tuple(out)

    # ---- store-free rule values
    stream_above_bytes = 32 << 30  # rules beyond this many bytes are summed on the fly instead of being cached

    def ptr_sum_supported(self, npt, fid):
        s = self.s
        if s.n > 4:  # generic-n kernels: resolvent traces of Hermitian series from the tridiagonal form; G, and series that are not Hermitian, from the inverse of every node
            if fid in (L.F_DOS, L.F_TRGLOC) and self.hermitian():
                return True
            return fid in (L.F_DOS, L.F_TRGLOC, L.F_GLOC)
        return (npt > 128 and self.hermitian() and
                not (fid in (L.F_LINEAR, L.F_LINEAR_X) and s.n != 1))

    def hermitian(self):
        """H_{-R} = H_R^dagger on the stored coefficient array (what the library detects at upload); cached per
        coefficient generation."""
        cached = getattr(self, "_herm", None)
        if cached is not None and cached[0] == self.generation and cached[1] is self.s.c:
            return cached[2]
        v = self._hermitian_now()
        self._herm = (self.generation, self.s.c, v)
        return v

    def _hermitian_now(self):
        c = self.s.c
        flip = c[tuple(slice(None, None, -1) for _ in range(self.s.d))]
        sym = all(2 * f + m - 1 == 0 for f, m in zip(np.atleast_1d(self.s.first), c.shape[:self.s.d]))
        return bool(sym and np.array_equal(c, np.conj(np.swapaxes(flip, -1, -2))))

    def ptr_sum(self, npt, fid, params=(), sweep=None, nsyms=1):
        """rule(f, B) on the full npt^d grid without materialising H(k) (abz_ptr_sum); with `kshard` set, this
        rank's slab followed by the all-reduce.  Returns complex [n_sweep, ncomp] like DeviceRule.reduce."""
        s = self.s
        ncomp = {L.F_GLOC: s.n * s.n, L.F_LINEAR_X: s.d}.get(fid, 1)
        params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
        swept = fid in (L.F_DOS, L.F_TRGLOC, L.F_GLOC, L.F_DOS_EIG)
        if swept:
            sw = np.ascontiguousarray(np.asarray(sweep, dtype=np.float64).reshape(-1))
            ns, psw = len(sw), sw.ctypes.data_as(L.c_f64p)
        else:
            ns, psw = 1, None
        out = np.zeros((ns, ncomp, 2))
        z0, z1 = 0, int(npt)
        if self.kshard and self.kshard[1] > 1:
            z0, z1 = slab_range(int(npt), *self.kshard)
        if z1 > z0:
            L.check(L.lib().abz_ptr_sum(self.h, int(npt), z0, z1, fid, params.ctypes.data_as(L.c_f64p) if len(params) else None,
                                        len(params), psw, ns, int(nsyms), out.ctypes.data_as(L.c_f64p)))
        if self.kshard and self.kshard[1] > 1:
            out = self.allreduce(out)
        return out.view(np.complex128).reshape(ns, ncomp)

    # ---- cached PTR rules
    def rule(self, npt, syms=None, want=L.WANT_H):
        """Cached rule.  With `self.kshard = (rank, world)` set (dist.kshard) the rule holds this rank's
        share of the nodes only -- a slab of the outermost variable of a full grid, or every world-th
        irreducible node -- and its reductions are summed over the ranks by `self.allreduce`."""
        # rules of a Hermitian series keep H(k) as its upper triangle (n^2 planes instead of 2 n^2: every built-in
        # integrand reads those planes only, export() still returns full matrices); ABZ_RULE_COMPACT=0: the reference's
        # full SMatrix layout
        want = self._layout_want(want)
        key = (int(npt), _syms_key(syms), int(want), self.kshard)
        r = self.rules.pop(key, None)
        if r is None:
            # a cached superset also serves
            for (n2, s2, w2, k2), r2 in list(self.rules.items()):
                if n2 == key[0] and s2 == key[1] and self._serves(w2, want) and k2 == self.kshard:
                    r = self.rules.pop((n2, s2, w2, k2))
                    key = (n2, s2, w2, k2)
                    break
        if r is None:
            r = DeviceRule(self, npt, syms, want)
            self.rule_bytes += r.nbytes
            while self.rule_bytes > self.max_rule_bytes and self.rules:
                old_key = next(iter(self.rules))
                old = self.rules.pop(old_key)
                self.rule_bytes -= old.nbytes
                old.close()
        self.rules[key] = r  # most recently used last
        return r

    def has_rule(self, npt, syms=None, want=L.WANT_H):
        """Is a rule serving this request already resident?"""
        k0, k1, want = int(npt), _syms_key(syms), self._layout_want(want)
        return any(n2 == k0 and s2 == k1 and self._serves(w2, want) and k2 == self.kshard for (n2, s2, w2, k2) in self.rules)

    def _layout_want(self, want):
        # (1...16 bands; the library ignores the bit where it has no upper-triangle kernel and reports the layout it built)
        if (want & L.WANT_H) and self.s.n <= 16 and os.environ.get("ABZ_RULE_COMPACT", "1") != "0" and self.hermitian():
            want |= L.WANT_H_COMPACT
        return int(want)

    @staticmethod
    def _serves(have, want):
        """A cached rule with planes `have` serves a request for `want`: every requested plane family is there and, when
        H planes are requested, in the requested layout (full SMatrix order or upper triangle) -- a zero-copy client of
        values_ptr must not be handed the other one."""
        return (have & want) == want and (not (want & L.WANT_H) or (have & L.WANT_H_COMPACT) == (want & L.WANT_H_COMPACT))

    def drop_rules(self):
        for r in self.rules.values():
            r.close()
        self.rules.clear()
        self.rule_bytes = 0
        if self._h is not None:  # ... and the rules the library keeps for its whole-solve entry points
            L.check(L.lib().abz_series_drop_rules(self.h))


def slab_range(n, rank, world):
    """Balanced contiguous share [a, b) of range(n) for `rank` of `world`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _syms_key(syms):
    if syms is None:
        return None
    return np.ascontiguousarray(np.rint(np.asarray(syms)).astype(np.int32)).tobytes()


def symptr_rule(npt, d, syms, ctx=None):
    """Irreducible grid nodes (0-based indices, column-major order) and integer weights.
    ref: AutoSymPTR.symptr_rule as called at src/fourier.jl:271.  With a device context the orbit
    tables are computed on the GPU (abz_symptr_rule_device), otherwise by the host routine; both give
    bit-identical integers."""
    S = np.ascontiguousarray(np.rint(np.asarray(syms)).astype(np.int32).reshape(-1, d, d))
    if not np.allclose(S, np.asarray(syms).reshape(-1, d, d)):
        raise ValueError("symmetries must be integer matrices in the lattice basis")
    _, pS = L.i32(S)
    n = C.c_int64(0)
    if ctx is not None and len(S) <= 48:
        fn = lambda *a: L.lib().abz_symptr_rule_device(ctx.h, *a)
    else:
        fn = L.lib().abz_symptr_rule
    L.check(fn(npt, d, pS, len(S), C.byref(n), None, None))
    idx = np.empty((n.value, d), dtype=np.int32)
    w = np.empty(n.value, dtype=np.int64)
    L.check(fn(npt, d, pS, len(S), C.byref(n), idx.ctypes.data_as(L.c_i32p), w.ctypes.data_as(L.c_i64p)))
    return idx, w


class DeviceRule:
    """abz_rule handle: FourierPTR (syms None) or FourierMonkhorstPack values resident in HBM.
    ref: src/fourier.jl:127-174,210-277."""

    def __init__(self, dev: DeviceSeries, npt, syms, want):
        self.dev = dev
        self.npt = int(npt)
        self.want = int(want)
        self.syms = syms
        self.nsyms = 1 if syms is None else len(syms)
        h = C.c_void_p()
        d, n = dev.s.d, dev.s.n
        self.shard = dev.kshard
        rank, world = self.shard if self.shard else (0, 1)
        if syms is None:
            self.nk = self.npt ** d  # nodes of the whole rule (numevals counts these)
            if world == 1:
                L.check(L.lib().abz_ptr_rule_build(dev.h, self.npt, 0, None, None, want, C.byref(h)))
                self.nk_local = self.nk
            else:
                if d < 2:
                    raise ValueError("k-sharding needs at least two variables")
                z0, z1 = slab_range(self.npt, rank, world)
                self.nk_local = (z1 - z0) * self.npt ** (d - 1)
                if z1 > z0:
                    L.check(L.lib().abz_ptr_rule_build_slab(dev.h, self.npt, z0, z1, want, C.byref(h)))
        elif world == 1 and len(syms) <= 48 and os.environ.get("ABZ_SYM_DEVICE", "1") != "0":
            # orbit tables, contraction plan and values all on the device (abz_ptr_rule_build_sym): the node list
            # never visits the host
            S = np.ascontiguousarray(np.rint(np.asarray(syms)).astype(np.int32).reshape(-1, d, d))
            if not np.allclose(S, np.asarray(syms).reshape(-1, d, d)):
                raise ValueError("symmetries must be integer matrices in the lattice basis")
            L.check(L.lib().abz_ptr_rule_build_sym(dev.h, self.npt, S.ctypes.data_as(L.c_i32p), len(S), want, C.byref(h)))
            nk = C.c_int64(0)
            L.check(L.lib().abz_rule_info(h, C.byref(nk), None, None, None, None))
            self.nk = self.nk_local = int(nk.value)
        else:
            idx, w = symptr_rule(self.npt, d, syms, ctx=dev.ctx)
            self.nk = len(w)
            if world > 1:  # consecutive blocks keep the runs of shared outer coordinates together
                a, b = slab_range(len(w), rank, world)
                idx, w = np.ascontiguousarray(idx[a:b]), np.ascontiguousarray(w[a:b])
            self.nk_local = len(w)
            if len(w):
                L.check(L.lib().abz_ptr_rule_build(dev.h, self.npt, len(w), idx.ctypes.data_as(L.c_i32p),
                                                   w.ctypes.data_as(L.c_i64p), want, C.byref(h)))
        self._h = h if h.value else None
        self._closed = False
        self.generation = dev.generation
        if self._h is not None and want & L.WANT_H_COMPACT:  # the library drops the bit when the layout does not apply
            got = C.c_int(0)
            L.check(L.lib().abz_rule_info(self._h, None, None, None, None, C.byref(got)))
            self.want = want = int(got.value)
        per = ((n * n if want & L.WANT_H_COMPACT else 2 * n * n) if want & L.WANT_H else 0) + (n if want & (L.WANT_EIG | L.WANT_VEL) else 0) + \
              (d * n if want & L.WANT_VEL else 0)
        self.nbytes = 8 * per * self.nk_local
        self._fin = weakref.finalize(self, DeviceRule._destroy, self._h)

    @property
    def h(self):
        """The abz_rule handle (None for an empty k-shard); a stale rule is refilled first."""
        if self._closed:
            raise L.AbzError("DeviceRule was closed")
        if self._h is not None and self.generation != self.dev.generation:
            self.generation = self.dev.generation
            L.check(L.lib().abz_rule_rebuild(self._h))
        return self._h

    @staticmethod
    def _destroy(h):
        try:
            if h is not None:
                L.lib().abz_rule_destroy(h)
        except Exception:
            pass

    def _sum_over_ranks(self, a):
        """Partial sums of this rank's share -> the value of the whole rule (the all-reduce of a
        k-sharded solve)."""
        return self.dev.allreduce(a) if self.shard and self.shard[1] > 1 else a

    def close(self):
        self._fin()
        self._closed = True
        self._h = None

    def __len__(self):
        return self.nk

    def rebuild(self):
        """Re-evaluate all cached values in place from the series' current coefficients (async)."""
        if self._closed:
            raise L.AbzError("DeviceRule was closed")
        if self._h is not None:
            self.generation = self.dev.generation
            L.check(L.lib().abz_rule_rebuild(self._h))

    def reduce(self, fid, params=(), sweep=None, nsyms=None):
        """(sum_k w_k f(k, H(k); sweep_i)) / (npt^d nsyms) for every sweep value -> complex array
        [n_sweep, ncomp].  ref: rule(f, B) = quadsum(...), src/fourier.jl:204-207,289-292."""
        s = self.dev.s
        ncomp = {L.F_GLOC: s.n * s.n, L.F_LINEAR_X: s.d}.get(fid, 1)
        params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
        swept = fid in (L.F_DOS, L.F_TRGLOC, L.F_GLOC, L.F_DOS_EIG)
        if swept:
            sw = np.ascontiguousarray(np.asarray(sweep, dtype=np.float64).reshape(-1))
            ns = len(sw)
            psw = sw.ctypes.data_as(L.c_f64p)
        else:
            ns, psw = 1, None
        out = np.zeros((ns, ncomp, 2))
        pp = params.ctypes.data_as(L.c_f64p) if len(params) else None
        if self.h is not None:
            L.check(L.lib().abz_rule_reduce(self.h, fid, pp, len(params), psw, ns,
                                            self.nsyms if nsyms is None else nsyms, out.ctypes.data_as(L.c_f64p)))
        return self._sum_over_ranks(out).view(np.complex128).reshape(ns, ncomp)

    def reduce_device(self, fid, params, sweep_ptr, n_sweep, out_ptr, nsyms=None):
        """abz_rule_reduce_device: `sweep_ptr` / `out_ptr` are raw device addresses (e.g. tensor.data_ptr()) of
        [n_sweep] and [n_sweep][ncomp][2] doubles; enqueues on the context's stream and returns.  The caller
        follows with its collective on the same stream (dist.py)."""
        params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
        pp = params.ctypes.data_as(L.c_f64p) if len(params) else None
        if self.h is not None:
            L.check(L.lib().abz_rule_reduce_device(self.h, fid, pp, len(params), C.c_void_p(int(sweep_ptr)), int(n_sweep),
                                                   self.nsyms if nsyms is None else nsyms, C.c_void_p(int(out_ptr))))

    def values_ptr(self):
        """(device address, bytes) of the rule's value block."""
        base = C.c_void_p()
        nb = C.c_int64(0)
        L.check(L.lib().abz_rule_values_ptr(self.h, C.byref(base), C.byref(nb)))
        return int(base.value or 0), int(nb.value)

    def export(self, x=True, w=True, H=False, eig=False, vel=False):
        """Host copies in the reference's layout: x [nk,d], w [nk], H [nk,n,n], eig [nk,n], vel [nk,d,n]."""
        s = self.dev.s
        nk, n, d = self.nk_local, s.n, s.d  # a k-sharded rule exports this rank's nodes
        X = np.empty((nk, d)) if x else None
        W = np.empty(nk) if w else None
        Hb = np.empty((nk, n * n, 2)) if H else None
        E = np.empty((nk, n)) if eig else None
        V = np.empty((nk, d, n)) if vel else None
        ptr = lambda a: a.ctypes.data_as(L.c_f64p) if a is not None else None
        if self.h is not None:
            L.check(L.lib().abz_rule_export(self.h, ptr(X), ptr(W), ptr(Hb), ptr(E), ptr(V)))
        out = {}
        if x:
            out["x"] = X
        if w:
            out["w"] = W
        if H:
            Hc = Hb.view(np.complex128).reshape(nk, n, n).transpose(0, 2, 1)
            out["H"] = Hc[:, 0, 0] if s.scalar else np.ascontiguousarray(Hc)
        if eig:
            out["eig"] = E
        if vel:
            out["vel"] = V
        return out

    def ggr(self, Es):
        """sum_k w_k sum_bands ggr_formula(1/(2 npt), E, e, v...).  ref: src/dos_ggr.jl:58-65."""
        Es = np.ascontiguousarray(np.asarray(Es, dtype=np.float64).reshape(-1))
        out = np.zeros(len(Es))
        if self.h is not None:
            L.check(L.lib().abz_rule_ggr(self.h, Es.ctypes.data_as(L.c_f64p), len(Es), out.ctypes.data_as(L.c_f64p)))
        return self._sum_over_ranks(out)
