"""hyper-greco-amd: MI355X-native GKR prover for the BFV secret-key-encryption circuit.

Python host mirror of the reference's `bfv-gkr` API for this path (same names and argument meaning):
`BfvEncrypt.new(k) / setup / get_inputs / prove` [REF bfv-gkr/src/sk_encryption_circuit.rs:300-460] and
`LassoNode.prove_claim_reduction` [REF lasso/src/lasso.rs:57-114], over the C ABI of `include/hg.h`
(libhypergreco.so: hand-written HIP kernels for gfx950). There is no CPU fallback: without a HIP
device `Context()` raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HG_LIB") or os.path.join(_HERE, "libhypergreco.so")  # HG_LIB: an experimental build (scripts/build_variant.sh)
HG_MAX_K = 16
P = 0xFFFFFFFF00000001

u64p = C.POINTER(C.c_uint64)


class HgParams(C.Structure):
    _fields_ = [("n", C.c_uint32), ("k", C.c_uint32), ("s_bound", C.c_uint64), ("e_bound", C.c_uint64),
                ("k1_bound", C.c_uint64), ("r1_bounds", C.c_uint64 * HG_MAX_K), ("r2_bounds", C.c_uint64 * HG_MAX_K),
                ("qis", C.c_uint64 * HG_MAX_K), ("k0is", C.c_uint64 * HG_MAX_K)]


class HgTimings(C.Structure):
    _fields_ = [("witness_ms", C.c_double), ("upload_ms", C.c_double), ("prove_ms", C.c_double), ("gpu_ms", C.c_double),
                ("total_ms", C.c_double), ("enqueue_ms", C.c_double), ("sync_ms", C.c_double), ("replay_ms", C.c_double)]


class HgKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("algo_bytes", C.c_double), ("model_bytes", C.c_double), ("hbm_bytes", C.c_double)]


EXPORTS = [
    "hg_last_error", "hg_device_count", "hg_create", "hg_destroy", "hg_set_option", "hg_params_builtin", "hg_params_derive", "hg_grand_product", "hg_fold", "hg_setup", "hg_pk_free",
    "hg_pk_lasso_layout", "hg_pk_info", "hg_pk_node_eq_form", "hg_witness_from_json", "hg_witness_synthetic", "hg_witness_from_arrays",
    "hg_witness_get", "hg_witness_free", "hg_prove", "hg_warmup", "hg_prove_stream", "hg_verify", "hg_verify_device", "hg_prove_mode", "hg_prove_resident_mode", "hg_verify_mode", "hg_group_local", "hg_group_external", "hg_group_free", "hg_prove_resident_mode_sharded", "hg_witness_gen", "hg_witness_gen_into", "hg_witness_gen_shard", "hg_values_info", "hg_values_peak_bytes", "hg_values_free", "hg_values_get", "hg_comm_unique_id", "hg_comm_init", "hg_comm_destroy", "hg_comm_count", "hg_comm_selftest", "hg_prove_sharded", "hg_prove_shard_begin", "hg_prove_shard_combine", "hg_prove_shard_finish", "hg_shard_combine_host", "hg_prove_resident", "hg_circuit_eval", "hg_lasso_prove", "hg_lasso_prove_at", "hg_lasso_num_challenges", "hg_sumcheck", "hg_mle_eval",
    "hg_ntt", "hg_challenges", "hg_challenges_bn254", "hg_bn254_field_op", "hg_sumcheck_bn254", "hg_grand_product_bn254", "hg_lasso_prove_bn254", "hg_witness_from_json_bn254", "hg_circuit_eval_bn254", "hg_prove_bn254", "hg_verify_bn254", "hg_mle_eval_bn254", "hg_ntt_bn254", "hg_profile", "hg_profile_select", "hg_profile_reset", "hg_profile_get",
]


def build(force=False):
    """Compiles csrc/ for gfx950 into libhypergreco.so (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j4"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run __graft_entry__.build() (the HIP extension is required)")
        L = C.CDLL(_LIB_PATH)
        L.hg_last_error.restype = C.c_char_p
        L.hg_create.restype = C.c_void_p
        L.hg_create.argtypes = [C.c_int]
        L.hg_destroy.argtypes = [C.c_void_p]
        L.hg_setup.argtypes = [C.c_void_p, C.POINTER(HgParams), C.POINTER(C.c_void_p)]
        L.hg_pk_free.argtypes = [C.c_void_p]
        L.hg_pk_lasso_layout.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.hg_pk_info.argtypes = [C.c_void_p, u64p]
        L.hg_witness_from_json.argtypes = [C.POINTER(HgParams), C.c_char_p, C.POINTER(C.c_void_p)]
        L.hg_witness_synthetic.argtypes = [C.POINTER(HgParams), C.c_uint64, C.POINTER(C.c_void_p)]
        L.hg_witness_from_arrays.argtypes = [C.POINTER(HgParams)] + [u64p] * 7 + [C.POINTER(C.c_void_p)]
        L.hg_witness_get.restype = C.c_int64
        L.hg_witness_get.argtypes = [C.c_void_p, C.c_int, u64p, C.c_size_t]
        L.hg_witness_free.argtypes = [C.c_void_p]
        L.hg_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
        L.hg_witness_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(HgTimings)]
        L.hg_witness_gen_into.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HgTimings)]
        L.hg_values_free.argtypes = [C.c_void_p]
        L.hg_values_get.restype = C.c_int64
        L.hg_values_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_size_t]
        L.hg_prove_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
        L.hg_prove_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(u64p), C.POINTER(C.c_size_t)]
        L.hg_prove_shard_combine.argtypes = [C.c_void_p, u64p, C.c_int, C.c_size_t]
        L.hg_prove_shard_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
        L.hg_verify.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.hg_verify_bn254.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.hg_circuit_eval.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t]
        L.hg_lasso_prove.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p]
        L.hg_lasso_prove_at.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p]
        L.hg_sumcheck.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(u64p), C.POINTER(C.c_int), u64p, C.c_size_t,
                                  u64p, C.c_size_t, u64p, u64p, u64p, u64p]
        L.hg_mle_eval.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p, u64p]
        L.hg_ntt.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int, C.c_size_t, u64p]
        L.hg_challenges.argtypes = [C.c_size_t, u64p]
        L.hg_challenges_bn254.argtypes = [C.c_size_t, u64p]
        L.hg_bn254_field_op.argtypes = [C.c_void_p, C.c_int, C.c_size_t, u64p, u64p, u64p]
        L.hg_sumcheck_bn254.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(u64p), u64p, C.c_size_t, u64p, C.c_size_t,
                                        u64p, u64p, u64p, u64p]
        L.hg_grand_product_bn254.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(u64p), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t,
                                             C.POINTER(C.c_size_t), u64p, u64p]
        L.hg_lasso_prove_bn254.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t), u64p]
        L.hg_mle_eval_bn254.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p, u64p]
        L.hg_witness_from_json_bn254.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.hg_circuit_eval_bn254.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, u64p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.hg_prove_bn254.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        L.hg_ntt_bn254.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int, C.c_size_t, u64p]
        L.hg_profile.argtypes = [C.c_void_p, C.c_int]
        L.hg_profile_select.argtypes = [C.c_void_p, C.c_char_p]
        L.hg_profile_reset.argtypes = [C.c_void_p]
        L.hg_profile_get.argtypes = [C.c_void_p, C.POINTER(HgKernelStat), C.c_int]
        _lib = L
    return _lib


class HgError(RuntimeError):
    pass


def _check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise HgError(lib().hg_last_error().decode())
    return rc


def _ptr(a):
    return a.ctypes.data_as(u64p)


def params_builtin(n, k):
    p = HgParams()
    _check(lib().hg_params_builtin(n, k, C.byref(p)))
    return p


def challenges(n):
    out = np.zeros(n, dtype=np.uint64)
    _check(lib().hg_challenges(n, _ptr(out)))
    return out


class Context:
    """One per GPU (hg_ctx): HIP stream, workspace arena, challenge chain in HBM."""

    def __init__(self, device=0):
        h = lib().hg_create(device)
        if not h:
            raise HgError(lib().hg_last_error().decode())
        self.h = C.c_void_p(h)

    def close(self):
        if self.h:
            lib().hg_destroy(self.h)
            self.h = None

    def set_option(self, name, value):
        """hg_set_option: "one_stream" (1 = no cross-stream overlap, for per-kernel timings)."""
        L = lib()
        L.hg_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        if L.hg_set_option(self.h, name.encode(), int(value)) != 0:
            raise HgError(L.hg_last_error().decode())

    def profile(self, level):
        lib().hg_profile(self.h, level)

    def profile_select(self, name):
        _check(lib().hg_profile_select(self.h, name.encode()))

    def profile_reset(self):
        lib().hg_profile_reset(self.h)

    def profile_get(self):
        arr = (HgKernelStat * 32)()
        n = lib().hg_profile_get(self.h, arr, 32)
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=arr[i].total_ms, algo_bytes=arr[i].algo_bytes, model_bytes=arr[i].model_bytes, hbm_bytes=arr[i].hbm_bytes)
                for i in range(n)]

    # ---- BN254 slice: elements are Python ints at this level, 4 little-endian u64 limbs at the C ABI
    @staticmethod
    def _fr_pack(vals):
        a = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            for k in range(4):
                a[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        return a.reshape(-1)

    @staticmethod
    def _fr_unpack(a):
        a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
        return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]

    def bn254_field_op(self, op, a, b):
        pa, pb = self._fr_pack(a), self._fr_pack(b)
        out = np.zeros_like(pa)
        _check(lib().hg_bn254_field_op(self.h, op, len(a), _ptr(pa), _ptr(pb), _ptr(out)))
        return self._fr_unpack(out)

    def grand_product_bn254(self, tables, chain_skip=0):
        """prove_grand_product over bn256::Fr: (proof bytes, final claims, point)."""
        nb, ln = len(tables), len(tables[0])
        nv = ln.bit_length() - 1
        packed = [self._fr_pack(t) for t in tables]
        ptrs = (u64p * nb)(*[_ptr(t) for t in packed])
        cap = 32 * (nb + sum(4 * n + 2 * nb for n in range(nv)) + 2 * nb + 64)
        buf = (C.c_uint8 * cap)()
        ln_out = C.c_size_t(0)
        claims = np.zeros(nb * 4, dtype=np.uint64)
        point = np.zeros(max(nv, 1) * 4, dtype=np.uint64)
        _check(lib().hg_grand_product_bn254(self.h, nb, ln, ptrs, chain_skip, buf, cap, C.byref(ln_out), _ptr(claims), _ptr(point)))
        return C.string_at(buf, ln_out.value), self._fr_unpack(claims), self._fr_unpack(point)[:nv]

    def lasso_prove_bn254(self, pk, lasso_in, chain_skip=0, cap=1 << 22):
        """LassoNode::prove_claim_reduction over bn256::Fr on a table of small integers: (proof bytes, r, claimed sum)."""
        n = len(lasso_in)
        nu = n.bit_length() - 1
        packed = self._fr_pack(lasso_in)
        buf = (C.c_uint8 * cap)()
        ln = C.c_size_t(0)
        claim = np.zeros((nu + 1) * 4, dtype=np.uint64)
        _check(lib().hg_lasso_prove_bn254(self.h, pk.h, _ptr(packed), chain_skip, buf, cap, C.byref(ln), _ptr(claim)))
        vals = self._fr_unpack(claim)
        return C.string_at(buf, ln.value), vals[:nu], vals[nu]

    def circuit_eval_bn254(self, pk, witness, which):
        """Circuit::evaluate over bn256::Fr: which = 0 the sum node, 1 the Lasso input node, 2 the ct0is table (Python ints)."""
        n = C.c_size_t(0)
        cap = 1 << (pk.params.n.bit_length() + 6)
        out = np.zeros(cap * 4, dtype=np.uint64)
        _check(lib().hg_circuit_eval_bn254(self.h, pk.h, witness.h, which, _ptr(out), cap, C.byref(n)))
        return self._fr_unpack(out[:n.value * 4])

    def prove_bn254(self, pk, witness, cap=1 << 24):
        """BfvEncrypt::prove over bn256::Fr: (proof bytes, witness-generation ms, prove ms)."""
        buf = (C.c_uint8 * cap)()
        ln = C.c_size_t(0)
        ms = (C.c_double * 2)()
        _check(lib().hg_prove_bn254(self.h, pk.h, witness.h, buf, cap, C.byref(ln), ms))
        return C.string_at(buf, ln.value), ms[0], ms[1]

    def mle_eval_bn254(self, table, point):
        nv = (len(table) - 1).bit_length()
        pt, pp = self._fr_pack(table), self._fr_pack(point) if len(point) else np.zeros(4, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().hg_mle_eval_bn254(self.h, _ptr(pt), nv, _ptr(pp), _ptr(out)))
        return self._fr_unpack(out)[0]

    def ntt_bn254(self, rows, inverse=False):
        """rows: list of equal-length lists (a batch of transforms)."""
        n = len(rows[0])
        flat = self._fr_pack([v for r in rows for v in r])
        out = np.zeros_like(flat)
        _check(lib().hg_ntt_bn254(self.h, _ptr(flat), n.bit_length() - 1, 1 if inverse else 0, len(rows), _ptr(out)))
        vals = self._fr_unpack(out)
        return [vals[i * n:(i + 1) * n] for i in range(len(rows))]

    def sumcheck_bn254(self, kind, tables, pw, claim, chain_skip=0):
        """gkr::sum_check::prove_sum_check over bn256::Fr: (msgs, point, evals, sums) as lists of ints."""
        ntab = len(tables)
        nv = (len(tables[0]) - 1).bit_length()
        d = 3 if kind == 1 else 2
        packed = [self._fr_pack(t) for t in tables]
        ptrs = (u64p * ntab)(*[_ptr(t) for t in packed])
        ppw = self._fr_pack(pw) if len(pw) else np.zeros(4, dtype=np.uint64)
        pcl = self._fr_pack([claim])
        msgs = np.zeros(max(nv * (d + 1), 1) * 4, dtype=np.uint64)
        point = np.zeros(max(nv, 1) * 4, dtype=np.uint64)
        evals = np.zeros(ntab * 4, dtype=np.uint64)
        sums = np.zeros(max(nv * d, 1) * 4, dtype=np.uint64)
        _check(lib().hg_sumcheck_bn254(self.h, kind, nv, ntab, ptrs, _ptr(ppw), len(pw), _ptr(pcl), chain_skip, _ptr(msgs), _ptr(point),
                                       _ptr(evals), _ptr(sums)))
        m = self._fr_unpack(msgs)[: nv * (d + 1)]
        return ([m[i * (d + 1):(i + 1) * (d + 1)] for i in range(nv)], self._fr_unpack(point)[:nv], self._fr_unpack(evals),
                [self._fr_unpack(sums)[i * d:(i + 1) * d] for i in range(nv)])

    # kernel-level entry points (parity tests)
    def sumcheck(self, kind, tables, is_base, pw, claim, chain_skip=0):
        ntab = len(tables)
        nv = int(np.log2(tables[0].size if is_base[0] else tables[0].size // 2))
        d = 3 if kind == 1 else 2
        tabs = [np.ascontiguousarray(t, dtype=np.uint64) for t in tables]
        ptrs = (u64p * ntab)(*[_ptr(t) for t in tabs])
        flags = (C.c_int * ntab)(*[int(b) for b in is_base])
        pw = np.ascontiguousarray(pw, dtype=np.uint64).reshape(-1)
        claim = np.ascontiguousarray(claim, dtype=np.uint64)
        msgs = np.zeros(nv * (d + 1) * 2, dtype=np.uint64)
        point = np.zeros(nv * 2, dtype=np.uint64)
        evals = np.zeros(ntab * 2, dtype=np.uint64)
        sums = np.zeros(nv * d * 2, dtype=np.uint64)
        _check(lib().hg_sumcheck(self.h, kind, nv, ntab, ptrs, flags, _ptr(pw), pw.size // 2, _ptr(claim), chain_skip,
                                 _ptr(msgs), _ptr(point), _ptr(evals), _ptr(sums)))
        return msgs, point, evals, sums

    def mle_eval(self, table, point):
        table = np.ascontiguousarray(table, dtype=np.uint64)
        point = np.ascontiguousarray(point, dtype=np.uint64)
        out = np.zeros(2, dtype=np.uint64)
        _check(lib().hg_mle_eval(self.h, _ptr(table), point.size // 2, _ptr(point), _ptr(out)))
        return out

    def ntt(self, data, log2n, inverse=False, batch=1):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros_like(data)
        _check(lib().hg_ntt(self.h, _ptr(data), log2n, int(inverse), batch, _ptr(out)))
        return out


class Witness:
    """BfvSkEncryptArgs after get_inputs [REF sk_encryption_circuit.rs:64-73, 365-415]."""
    FIELDS = ["s", "e", "k1", "ais", "r1is", "r2is", "ct0is"]

    def __init__(self, handle, params):
        self.h = handle
        self.params = params

    @classmethod
    def from_json(cls, params, path):
        h = C.c_void_p()
        _check(lib().hg_witness_from_json(C.byref(params), path.encode(), C.byref(h)))
        return cls(h, params)

    @classmethod
    def from_json_bn254(cls, params, path):
        """One of the reference's bn254 fixtures (elements of bn256::Fr holding small signed integers)."""
        h = C.c_void_p()
        _check(lib().hg_witness_from_json_bn254(C.byref(params), path.encode(), C.byref(h)))
        return cls(h, params)

    @classmethod
    def synthetic(cls, params, seed):
        h = C.c_void_p()
        _check(lib().hg_witness_synthetic(C.byref(params), seed, C.byref(h)))
        return cls(h, params)

    @classmethod
    def from_arrays(cls, params, d):
        h = C.c_void_p()
        arrs = [np.ascontiguousarray(d[f], dtype=np.uint64) for f in cls.FIELDS]
        _check(lib().hg_witness_from_arrays(C.byref(params), *[_ptr(a) for a in arrs], C.byref(h)))
        return cls(h, params)

    def arrays(self):
        out = {}
        for i, f in enumerate(self.FIELDS):
            n = lib().hg_witness_get(self.h, i, None, 0)
            a = np.zeros(n, dtype=np.uint64)
            lib().hg_witness_get(self.h, i, _ptr(a), n)
            out[f] = a
        return out

    def __del__(self):
        try:
            if self.h:
                lib().hg_witness_free(self.h)
                self.h = None
        except Exception:
            pass


class ProverKey:
    def __init__(self, handle, params):
        self.h = handle
        self.params = params
        info = (C.c_uint64 * 6)()
        lib().hg_pk_info(self.h, info)
        self.nu, self.num_nodes, self.rows, self.alpha, self.lasso_in_id, self.sum_id = (int(x) for x in info)

    def lasso_layout(self):
        buf = C.create_string_buffer(1 << 16)
        _check(lib().hg_pk_lasso_layout(self.h, buf, 1 << 16))
        mems, lk = buf.value.decode().split("|")
        return mems.split(","), lk.split(";")

    def node_eq_form(self, node):
        """hg_pk_node_eq_form: how setup classified a node - dict(kind, eq_form, block_log2, window, terms, in_log2)."""
        out = (C.c_int64 * 6)()
        L = lib()
        L.hg_pk_node_eq_form.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        _check(L.hg_pk_node_eq_form(self.h, int(node), out))
        return {"kind": ("input", "vanilla", "fft", "lasso")[int(out[0])], "eq_form": bool(out[1]), "block_log2": int(out[2]), "window": int(out[3]),
                "terms": int(out[4]), "in_log2": int(out[5])}

    def circuit_eval(self, w):
        lasso_in = np.zeros(1 << self.nu, dtype=np.uint64)
        sum_out = np.zeros(self.params.k * 2 * self.params.n, dtype=np.uint64)
        _check(lib().hg_circuit_eval(self.h, w.h, _ptr(lasso_in), lasso_in.size, _ptr(sum_out), sum_out.size))
        return lasso_in, sum_out

    def free(self):
        if self.h:
            lib().hg_pk_free(self.h)
            self.h = None


class BfvEncrypt:
    """Mirror of `BfvEncrypt::<Params, K>` [REF sk_encryption_circuit.rs:300-523]."""

    def __init__(self, n, k=None):  # = BfvEncrypt::<SkEnc{n}_{k}x.._65537, k>::new(k); or BfvEncrypt(params) with a derived HgParams
        self.params = n if isinstance(n, HgParams) else params_builtin(n, k)

    @classmethod
    def new(cls, n, k):
        return cls(n, k)

    def setup(self, ctx):  # -> (pk); the verifier key of the reference is the same preprocessing
        h = C.c_void_p()
        _check(lib().hg_setup(ctx.h if ctx is not None else None, C.byref(self.params), C.byref(h)))
        return ProverKey(h, self.params)

    def warmup(self, ctx, pk):
        """hg_warmup: context-owned tables, witness staging and the recorded launch graph ahead of the first prove; returns the ms it took."""
        L = lib()
        L.hg_warmup.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.hg_warmup.restype = C.c_int
        ms = C.c_double(0)
        _check(L.hg_warmup(ctx.h, pk.h, C.byref(ms)))
        return ms.value

    def get_inputs(self, path):
        return Witness.from_json(self.params, path)

    def prove(self, ctx, pk, witness, cap=1 << 24, mode=0):
        """BfvEncrypt::prove; mode != 0: hg_prove_mode (bit 0 absorbing transcript, bit 1 extension-field memory checking)."""
        buf = (C.c_uint8 * cap)()
        ln = C.c_size_t(0)
        tm = HgTimings()
        L = lib()
        if mode:
            L.hg_prove_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
            _check(L.hg_prove_mode(ctx.h, pk.h, witness.h, mode, buf, cap, C.byref(ln), C.byref(tm)))
        else:
            _check(L.hg_prove(ctx.h, pk.h, witness.h, buf, cap, C.byref(ln), C.byref(tm)))
        return C.string_at(buf, ln.value), {f: getattr(tm, f) for f, _ in HgTimings._fields_}

    def prove_stream(self, ctx, pk, witnesses, cap_each=1 << 20):
        """hg_prove_stream: BfvEncrypt::prove for a run of witnesses, witness i+1's upload + evaluate under witness i's prove."""
        n = len(witnesses)
        L = lib()
        L.hg_prove_stream.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
        L.hg_prove_stream.restype = C.c_int
        hs = (C.c_void_p * max(n, 1))(*[w.h for w in witnesses])
        buf = (C.c_uint8 * (cap_each * max(n, 1)))()
        lens = (C.c_size_t * max(n, 1))()
        tm = HgTimings()
        _check(L.hg_prove_stream(ctx.h, pk.h, hs, n, buf, cap_each, lens, C.byref(tm)))
        raw = memoryview(buf)
        proofs = [bytes(raw[i * cap_each:i * cap_each + lens[i]]) for i in range(n)]
        return proofs, {f: getattr(tm, f) for f, _ in HgTimings._fields_}


class ResidentValues:
    """circuit.evaluate() output kept in HBM (hg_values)."""

    def __init__(self, handle, timings):
        self.h = handle
        self.timings = timings

    def node(self, ctx, node_id):
        n = lib().hg_values_get(ctx.h, self.h, node_id, None, 0)
        if n < 0:
            raise HgError(lib().hg_last_error().decode())
        a = np.zeros(n, dtype=np.uint64)
        lib().hg_values_get(ctx.h, self.h, node_id, _ptr(a), n)
        return a

    def info(self):
        """hg_values_info: resident bytes, bytes of the full table set, resident tables, tables."""
        out = (C.c_uint64 * 4)()
        L = lib()
        L.hg_values_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _check(L.hg_values_info(self.h, out))
        L.hg_values_peak_bytes.argtypes = [C.c_void_p]
        L.hg_values_peak_bytes.restype = C.c_int64
        return {"resident_bytes": int(out[0]), "full_bytes": int(out[1]), "resident_tables": int(out[2]), "tables": int(out[3]),
                "peak_bytes": int(L.hg_values_peak_bytes(self.h))}

    def free(self):
        if self.h:
            lib().hg_values_free(self.h)
            self.h = None


def witness_gen(ctx, pk, witness):
    h = C.c_void_p()
    tm = HgTimings()
    _check(lib().hg_witness_gen(ctx.h, pk.h, witness.h, C.byref(h), C.byref(tm)))
    return ResidentValues(h, {f: getattr(tm, f) for f, _ in HgTimings._fields_})


def witness_gen_shard(ctx, pk, witness, rank, world):
    """hg_witness_gen_shard: circuit.evaluate() keeping only the tables rank `rank` of a `world`-GPU proof reads."""
    h = C.c_void_p()
    tm = HgTimings()
    L = lib()
    L.hg_witness_gen_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(HgTimings)]
    _check(L.hg_witness_gen_shard(ctx.h, pk.h, witness.h, rank, world, C.byref(h), C.byref(tm)))
    return ResidentValues(h, {f: getattr(tm, f) for f, _ in HgTimings._fields_})


def witness_gen_into(ctx, pk, witness, values):
    """hg_witness_gen_into: circuit.evaluate() of another witness into the SAME resident tables (the launch graph recorded for
    `values` stays valid)."""
    tm = HgTimings()
    _check(lib().hg_witness_gen_into(ctx.h, pk.h, witness.h, values.h, C.byref(tm)))
    values.timings = {f: getattr(tm, f) for f, _ in HgTimings._fields_}
    return values


class ProofBuffer:
    """Reusable output buffer so the timed loop does no Python-side allocation."""

    def __init__(self, cap=1 << 22):
        self.cap = cap
        self.buf = (C.c_uint8 * cap)()
        self.len = C.c_size_t(0)
        self.tm = HgTimings()

    def bytes(self):
        return C.string_at(self.buf, self.len.value)

    def timings(self):
        return {f: getattr(self.tm, f) for f, _ in HgTimings._fields_}


def prove_resident(ctx, pk, values, out):
    _check(lib().hg_prove_resident(ctx.h, pk.h, values.h, out.buf, out.cap, C.byref(out.len), C.byref(out.tm)))
    return out


def prove_resident_mode(ctx, pk, values, out, mode):
    """hg_prove_resident_mode: the round-by-round prover of the f-4 protocol modes on resident node tables."""
    L = lib()
    L.hg_prove_resident_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
    _check(L.hg_prove_resident_mode(ctx.h, pk.h, values.h, mode, out.buf, out.cap, C.byref(out.len), C.byref(out.tm)))
    return out


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)


class Group:
    """hg_group: the ranks of a sharded round-by-round prove. Group.local(world): ranks are threads of this process. Group.external(fn,
    world): fn(words: numpy u64 view) adds the ranks' words lane-wise mod p IN PLACE (an all-gather + modular sum, say)."""

    def __init__(self, h, world, keep=None):
        self.h, self.world, self._keep = h, world, keep

    @classmethod
    def local(cls, world):
        L = lib()
        L.hg_group_local.restype = C.c_void_p
        L.hg_group_local.argtypes = [C.c_int]
        h = L.hg_group_local(world)
        if not h:
            raise HgError(lib().hg_last_error().decode())
        return cls(h, world)

    @classmethod
    def external(cls, fn, world):
        L = lib()

        def tramp(_user, words, n):
            try:
                fn(np.ctypeslib.as_array(words, shape=(n,)))
                return 0
            except Exception:   # the library turns a non-zero return into an error of the prove
                import traceback
                traceback.print_exc()
                return -1

        cb = REDUCE_FN(tramp)
        L.hg_group_external.restype = C.c_void_p
        L.hg_group_external.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        h = L.hg_group_external(C.cast(cb, C.c_void_p), None, world)
        if not h:
            raise HgError(lib().hg_last_error().decode())
        return cls(h, world, keep=cb)

    def __del__(self):
        try:
            if self.h:
                lib().hg_group_free.argtypes = [C.c_void_p]
                lib().hg_group_free(self.h)
                self.h = None
        except Exception:
            pass


def prove_resident_mode_sharded(ctx, pk, values, out, mode, rank, group):
    """hg_prove_resident_mode_sharded: this rank's run of the round-by-round prover, one all-reduce per sum-check round through `group`."""
    L = lib()
    L.hg_prove_resident_mode_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
    _check(L.hg_prove_resident_mode_sharded(ctx.h, pk.h, values.h, mode, rank, group.h, out.buf, out.cap, C.byref(out.len), C.byref(out.tm)))
    return out


def prove_shard_begin(ctx, pk, values, rank, world):
    """This rank's share of ONE proof; returns a numpy view (u64) of the partial result buffer, to be all-gathered
    across ranks and handed to prove_shard_combine before prove_shard_finish."""
    ptr = u64p()
    n = C.c_size_t(0)
    _check(lib().hg_prove_shard_begin(ctx.h, pk.h, values.h, rank, world, C.byref(ptr), C.byref(n)))
    return np.ctypeslib.as_array(ptr, shape=(n.value,))


def prove_shard_combine(ctx, gathered, world):
    """gathered: the ranks' partial buffers (world x n u64, rank-major); installs their lane-wise sum mod p."""
    gathered = np.ascontiguousarray(gathered, dtype=np.uint64).reshape(-1)
    _check(lib().hg_prove_shard_combine(ctx.h, _ptr(gathered), world, gathered.size // world))


def shard_combine_host(gathered):
    """hg_shard_combine_host: [world, n] canonical u64 lanes -> lane-wise sum mod p (host only, no device needed)."""
    a = np.ascontiguousarray(gathered, dtype=np.uint64)
    world, n = a.shape
    out = np.zeros(n, dtype=np.uint64)
    L = lib()
    L.hg_shard_combine_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    _check(L.hg_shard_combine_host(_ptr(a), world, n, _ptr(out)))
    return out


def prove_shard_finish(ctx, out):
    _check(lib().hg_prove_shard_finish(ctx.h, out.buf, out.cap, C.byref(out.len), C.byref(out.tm)))
    return out


def params_derive(n, k, qis, t=65537):
    """hg_params_derive: the constants emitter of scripts/circuit_sk.py:422-439 for ring degree n and moduli qis."""
    p = HgParams()
    arr = (C.c_uint64 * len(qis))(*qis)
    L = lib()
    L.hg_params_derive.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(HgParams)]
    _check(L.hg_params_derive(n, k, arr, t, C.byref(p)))
    return p


def grand_product(ctx, tables, chain_skip=0, cap=1 << 22):
    """hg_grand_product: prove_grand_product on base-field tables (numpy u64); -> (proof bytes, claims (nb x 2), point (nv x 2))."""
    nb, ln = len(tables), tables[0].size
    nv = ln.bit_length() - 1
    tabs = [np.ascontiguousarray(t, dtype=np.uint64) for t in tables]
    ptrs = (u64p * nb)(*[_ptr(t) for t in tabs])
    buf = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    claims = np.zeros(2 * nb, dtype=np.uint64)
    point = np.zeros(2 * nv, dtype=np.uint64)
    L = lib()
    L.hg_grand_product.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p, u64p]
    _check(L.hg_grand_product(ctx.h, nb, ln, ptrs, chain_skip, buf, cap, C.byref(n), _ptr(claims), _ptr(point)))
    return C.string_at(buf, n.value), claims.reshape(nb, 2), point.reshape(nv, 2)


def fold(ctx, table, is_base, r):
    """hg_fold: fix_var on the lowest variable; table: numpy u64 (2^nv base values or 2^nv (c0, c1) pairs flattened)."""
    t = np.ascontiguousarray(table, dtype=np.uint64)
    n_el = t.size if is_base else t.size // 2
    nv = n_el.bit_length() - 1
    out = np.zeros(n_el, dtype=np.uint64)  # 2^(nv-1) pairs
    rr = (C.c_uint64 * 2)(int(r[0]), int(r[1]))
    L = lib()
    L.hg_fold.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), u64p]
    _check(L.hg_fold(ctx.h, _ptr(t), nv, 1 if is_base else 0, rr, _ptr(out)))
    return out.reshape(-1, 2)


def device_count():
    """hg_device_count: HIP devices visible to this process (0 without a GPU)."""
    return int(lib().hg_device_count())


def comm_unique_id():
    """hg_comm_unique_id: the 128-byte RCCL id rank 0 shares with the other ranks."""
    buf = (C.c_uint8 * 128)()
    _check(lib().hg_comm_unique_id(buf))
    return bytes(buf)


def comm_init(ctx, uid, rank, world):
    """hg_comm_init (collective): RCCL communicator of this rank's context."""
    buf = (C.c_uint8 * 128).from_buffer_copy(uid)
    L = lib()
    L.hg_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    _check(L.hg_comm_init(ctx.h, buf, rank, world))


def comm_count(ctx):
    """hg_comm_count: ranks of the context's RCCL communicator (ncclCommCount)."""
    L = lib()
    n = C.c_int(0)
    L.hg_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    _check(L.hg_comm_count(ctx.h, C.byref(n)))
    return n.value


def comm_selftest(ctx, rank_buffers):
    """hg_comm_selftest: rank_buffers [world, n] u64 canonical lanes -> lane-wise sum mod p through the split / combine kernels."""
    a = np.ascontiguousarray(rank_buffers, dtype=np.uint64)
    world, n = a.shape
    out = np.zeros(n, dtype=np.uint64)
    L = lib()
    L.hg_comm_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    _check(L.hg_comm_selftest(ctx.h, _ptr(a), world, n, _ptr(out)))
    return out


def comm_destroy(ctx):
    L = lib()
    L.hg_comm_destroy.argtypes = [C.c_void_p]
    _check(L.hg_comm_destroy(ctx.h))


def prove_sharded(ctx, pk, values, out):
    """hg_prove_sharded (collective): this rank's share of ONE proof + the RCCL all-reduce inside the library + replay."""
    L = lib()
    L.hg_prove_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(HgTimings)]
    _check(L.hg_prove_sharded(ctx.h, pk.h, values.h, out.buf, out.cap, C.byref(out.len), C.byref(out.tm)))
    return out


def challenges_bn254(n):
    """First n Fiat-Shamir challenges over bn256::Fr (host)."""
    out = np.zeros(n * 4, dtype=np.uint64)
    _check(lib().hg_challenges_bn254(n, _ptr(out)))
    return Context._fr_unpack(out)


def verify(pk, witness, proof, mode=0):
    """BfvEncrypt::verify [REF sk_encryption_circuit.rs:462-517]: (accepted, reason). mode: see hg_verify_mode."""
    L = lib()
    L.hg_verify_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    rc = L.hg_verify_mode(pk.h, witness.h, mode, proof, len(proof)) if mode else L.hg_verify(pk.h, witness.h, proof, len(proof))
    if rc < 0:
        raise HgError(lib().hg_last_error().decode())
    return rc == 0, ("" if rc == 0 else lib().hg_last_error().decode())


def verify_device(ctx, pk, witness, proof):
    """hg_verify_device: BfvEncrypt::verify with the table-sized work on the device: (accepted, reason)."""
    L = lib()
    L.hg_verify_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    rc = L.hg_verify_device(ctx.h, pk.h, witness.h, proof, len(proof))
    if rc < 0:
        raise HgError(lib().hg_last_error().decode())
    return rc == 0, ("" if rc == 0 else lib().hg_last_error().decode())


def verify_bn254(pk, witness, proof):
    """BfvEncrypt::verify over bn256::Fr [REF sk_encryption_circuit.rs:462-517, 614-626]: (accepted, reason)."""
    rc = lib().hg_verify_bn254(pk.h, witness.h, proof, len(proof))
    if rc < 0:
        raise HgError(lib().hg_last_error().decode())
    return rc == 0, ("" if rc == 0 else lib().hg_last_error().decode())


class LassoNode:
    """Mirror of `LassoNode<F, E, 4, 65536>` as a gkr Node [REF lasso/src/lasso.rs:32-140]."""

    def __init__(self, pk):
        self.pk = pk

    def prove_claim_reduction(self, ctx, inputs, cap=1 << 24, chain_skip=0):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        assert inputs.size == 1 << self.pk.nu
        buf = (C.c_uint8 * cap)()
        ln = C.c_size_t(0)
        claim = np.zeros(2 * self.pk.nu + 2, dtype=np.uint64)
        _check(lib().hg_lasso_prove_at(ctx.h, self.pk.h, _ptr(inputs), chain_skip, buf, cap, C.byref(ln), _ptr(claim)))
        return bytes(buf[:ln.value]), claim
