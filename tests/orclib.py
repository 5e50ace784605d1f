"""ctypes bindings to oracle/liboracle.so — the CPU oracle (test infrastructure only)."""
import os as _os
# Two OpenMP runtimes share a test process (libgomp behind the oracle, libomp behind the product library); with the default
# active wait policy their idle teams spin against each other and the oracle gets SLOWER with more threads. Read at load time.
_os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
GOLDEN = os.path.join(ROOT, "tests", "golden")

u64p = C.POINTER(C.c_uint64)


class OrcParams(C.Structure):
    _fields_ = [("n", C.c_uint64), ("k", C.c_uint64), ("s_bound", C.c_uint64), ("e_bound", C.c_uint64),
                ("k1_bound", C.c_uint64), ("r1_bounds", u64p), ("r2_bounds", u64p), ("qis", u64p), ("k0is", u64p)]


class OrcInputs(C.Structure):
    _fields_ = [(f, u64p) for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = C.CDLL(so)
        _lib.orc_root_of_unity.restype = C.c_uint64
        _lib.orc_subtable_cutoff.restype = C.c_uint64
    return _lib


def ptr(a):
    return a.ctypes.data_as(u64p)


def constants(n, k):
    return json.load(open(os.path.join(GOLDEN, "constants.json")))[f"{n}_{k}"]


class Params:
    """Keeps the numpy arrays alive next to the ctypes struct."""

    def __init__(self, c):
        self.c = c
        self.n, self.k = c["n"], c["k"]
        self.L = self.n.bit_length()  # log2(n) + 1
        self.arrs = {f: np.array(c[f], dtype=np.uint64) for f in ("r1_bounds", "r2_bounds", "qis", "k0is")}
        self.struct = OrcParams(c["n"], c["k"], c["s_bound"], c["e_bound"], c["k1_bound"],
                                *(ptr(self.arrs[f]) for f in ("r1_bounds", "r2_bounds", "qis", "k0is")))


def params(n, k):
    return Params(constants(n, k))


def layout_inputs(n, k, w):
    """Python restatement of get_inputs / Poly::{new,new_padded,new_shifted}
    [REF sk_encryption_circuit.rs:365-415, poly.rs:12-44] for cross-checking the product's loader."""
    L = n.bit_length()
    SZ = 1 << L

    def arr(x):
        return np.array([int(v) for v in x], dtype=np.uint64)

    def padded(x):
        a = np.zeros(SZ, dtype=np.uint64)
        a[:len(x)] = arr(x)
        return a

    def shifted(x, size):
        pad = max(size - len(x), 0)
        v = np.concatenate([np.zeros(pad, dtype=np.uint64), arr(x)])
        npow = 1 << (size - 1).bit_length()
        out = np.zeros(npow, dtype=np.uint64)
        out[:len(v)] = v
        return out

    d = {}
    d["s"] = padded(w["s"])
    d["e"] = shifted(w["e"], SZ - 1)
    d["k1"] = shifted(w["k1"], SZ - 1)
    d["ais"] = np.concatenate([padded(w["ais"][z]) for z in range(k)])
    d["r1is"] = np.concatenate([padded(w["r1is"][z]) for z in range(k)])
    d["r2is"] = np.concatenate([np.concatenate([arr(w["r2is"][z]), np.zeros(1, dtype=np.uint64)]) for z in range(k)])
    ct = []
    for z in range(k):
        c = shifted(w["ct0is"][z], SZ)
        ct.append(np.concatenate([c[1:], np.zeros(1, dtype=np.uint64)]))
    d["ct0is"] = np.concatenate(ct)
    return d


class Inputs:
    def __init__(self, d):
        self.d = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in d.items()}
        self.struct = OrcInputs(*(ptr(self.d[f]) for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is")))


def fixture_inputs(n, k, bits):
    w = json.load(open(os.path.join(GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json")))
    return Inputs(layout_inputs(n, k, w))


def _err():
    return C.create_string_buffer(512)


def prove(p, inp, threads=1, cap=1 << 24):
    buf = (C.c_uint8 * cap)()
    ln = C.c_size_t(0)
    tm = (C.c_double * 2)()
    err = _err()
    rc = lib().orc_prove(C.byref(p.struct), C.byref(inp.struct), threads, buf, C.c_size_t(cap), C.byref(ln), tm, err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return bytes(buf[:ln.value]), (tm[0], tm[1])


def verify(p, inp, proof, threads=1):
    err = _err()
    rc = lib().orc_verify(C.byref(p.struct), C.byref(inp.struct), threads, proof, C.c_size_t(len(proof)), err, C.c_size_t(512))
    return rc == 0, err.value.decode()


def circuit_eval(p, inp):
    info = (C.c_uint64 * 3)()
    err = _err()
    lib().orc_circuit_eval(C.byref(p.struct), C.byref(inp.struct), None, C.c_size_t(0), None, info, err, C.c_size_t(512))
    nu = info[0]
    lasso_in = np.zeros(1 << nu, dtype=np.uint64)
    sum_out = np.zeros(p.k << p.L, dtype=np.uint64)
    rc = lib().orc_circuit_eval(C.byref(p.struct), C.byref(inp.struct), ptr(lasso_in), C.c_size_t(lasso_in.size), ptr(sum_out), info, err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return lasso_in, sum_out, dict(nu=int(info[0]), num_nodes=int(info[1]), rows=int(info[2]))


def lasso_prove(p, lasso_in, threads=1, cap=1 << 24):
    buf = (C.c_uint8 * cap)()
    ln = C.c_size_t(0)
    nu = int(np.log2(lasso_in.size))
    claim = np.zeros(2 * nu + 2, dtype=np.uint64)
    err = _err()
    rc = lib().orc_lasso_prove(C.byref(p.struct), ptr(lasso_in), threads, buf, C.c_size_t(cap), C.byref(ln), ptr(claim), err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return bytes(buf[:ln.value]), claim


def lasso_verify(p, proof):
    err = _err()
    rc = lib().orc_lasso_verify(C.byref(p.struct), proof, C.c_size_t(len(proof)), err, C.c_size_t(512))
    return rc == 0, err.value.decode()


def lasso_layout(p):
    buf = C.create_string_buffer(1 << 16)
    n = lib().orc_lasso_layout(C.byref(p.struct), buf, C.c_size_t(1 << 16))
    assert n > 0
    mems, lk = buf.value.decode().split("|")
    return mems.split(","), lk.split(";")


def keccak256(data: bytes) -> bytes:
    """Keccak-256 of the C oracle (pinned by the reference's chain KATs in test_oracle_kats.py)."""
    import ctypes
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    out = (ctypes.c_uint8 * 32)()
    lib().orc_keccak256(buf, ctypes.c_size_t(len(data)), out)
    return bytes(out)


def bn254():
    """The BN254 Python oracle (oracle/bn254.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_bn254", os.path.join(ROOT, "oracle", "bn254.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def bn254_gkr():
    """The BN254 whole-proof Python oracle (oracle/bn254_gkr.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_bn254_gkr", os.path.join(ROOT, "oracle", "bn254_gkr.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def bn254_lasso_fns(p):
    """(prove, verify) callbacks for bn254_gkr.prove / .verify: the Lasso node of oracle/bn254.py on the C oracle's integer tables."""
    import numpy as np
    bn = bn254()
    state = {}

    def prove_fn(vin, ch):
        P = lasso_polys(p, np.array(vin, dtype=np.uint64))
        state["P"] = P
        els, r, v = bn.lasso_prove(P, ch)
        return els, r, v, bn.lasso_challenge_count(P["nu"])

    def verify_fn(els, ch, layout=None):
        P = layout or state["P"]
        r, v, pos = bn.lasso_verify(els, P["nu"], P["mem_dim"], P["mem_cutoff"], ch, partial=True)
        return r, v, pos, bn.lasso_challenge_count(P["nu"])

    return prove_fn, verify_fn


def lasso_polys(p, lasso_in):
    """Integer tables of the Lasso node from the C oracle (orc_lasso_polys) as Python lists, plus lookup_mems."""
    import ctypes as C
    import numpy as np
    L = lib()
    nu, A, rows = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    err = C.create_string_buffer(256)
    lin = np.ascontiguousarray(lasso_in, dtype=np.uint64)
    u64p, u8p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    L.orc_lasso_polys.argtypes = [C.c_void_p, u64p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), u64p, u64p, u64p,
                                  u64p, u8p, i32p, u64p, C.c_char_p, C.c_size_t]
    pp = C.cast(C.byref(p.struct), C.c_void_p)
    assert L.orc_lasso_polys(pp, ptr(lin), C.byref(nu), C.byref(A), C.byref(rows), None, None, None, None, None, None, None, err, 256) == 0, err.value
    N = 1 << nu.value
    dims = np.zeros(4 * N, dtype=np.uint64); rd = np.zeros(A.value * N, dtype=np.uint64)
    fin = np.zeros(A.value * 65536, dtype=np.uint64); ep = np.zeros(A.value * N, dtype=np.uint64)
    rl = np.zeros(N, dtype=np.uint8); md = np.zeros(A.value, dtype=np.int32); mc = np.zeros(A.value, dtype=np.uint64)
    assert L.orc_lasso_polys(pp, ptr(lin), C.byref(nu), C.byref(A), C.byref(rows), ptr(dims), ptr(rd), ptr(fin), ptr(ep),
                             rl.ctypes.data_as(u8p), md.ctypes.data_as(i32p), ptr(mc), err, 256) == 0, err.value
    _, lookups = lasso_layout(p)
    lookup_mems = [[int(x) for x in l.split(":")[2].split("/")] for l in lookups]
    a = A.value
    return dict(nu=nu.value, A=a, rows=rows.value,
                dims=[dims[c * N:(c + 1) * N].tolist() for c in range(4)],
                read_cts=[rd[m * N:(m + 1) * N].tolist() for m in range(a)],
                final_cts=[fin[m * 65536:(m + 1) * 65536].tolist() for m in range(a)],
                e_polys=[ep[m * N:(m + 1) * N].tolist() for m in range(a)],
                row_lookup=rl.tolist(), mem_dim=md.tolist(), mem_cutoff=mc.tolist(), lookup_mems=lookup_mems)


# ---- the C++ oracle over bn256::Fr (oracle/fr.hpp: symbols orcbn_*) and the protocol modes of both field builds -------------
R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MODE_ABSORB, MODE_EXT_MEMCHECK = 1, 2


def _sym(field, name):
    return getattr(lib(), ("orcbn_" if field == "bn254" else "orc_") + name)


def limbs_of(field):
    return (4, 4) if field == "bn254" else (1, 2)


def to_limbs(vals, nl):
    """Python integers -> canonical little-endian u64 limbs (nl per element)."""
    a = np.zeros(len(vals) * nl, dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(nl):
            a[i * nl + j] = (int(v) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return a


def from_limbs(a, nl):
    a = [int(x) for x in a]
    return [sum(a[i * nl + j] << (64 * j) for j in range(nl)) for i in range(len(a) // nl)]


def f_binop(field, op, a, b):
    nl = limbs_of(field)[0]
    out = np.zeros(len(a) * nl, dtype=np.uint64)
    _sym(field, "f_binop")(op, C.c_size_t(len(a)), ptr(to_limbs(a, nl)), ptr(to_limbs(b, nl)), ptr(out))
    return from_limbs(out, nl)


def challenge_chain(field, n):
    nl = limbs_of(field)[0]
    out = np.zeros(n * nl, dtype=np.uint64)
    _sym(field, "challenge_chain")(C.c_size_t(n), ptr(out))
    return from_limbs(out, nl)


def wire_roundtrip(field, vals):
    nl = limbs_of(field)[0]
    buf = (C.c_uint8 * (len(vals) * 8 * nl))()
    back = np.zeros(len(vals) * nl, dtype=np.uint64)
    f = _sym(field, "wire_roundtrip")
    f.restype = C.c_size_t
    n = f(C.c_size_t(len(vals)), ptr(to_limbs(vals, nl)), buf, ptr(back))
    return bytes(buf[:n]), from_limbs(back, nl)


def root_of_unity_f(field, log2n):
    nl = limbs_of(field)[0]
    out = np.zeros(nl, dtype=np.uint64)
    _sym(field, "root_of_unity_limbs")(C.c_size_t(log2n), ptr(out))
    return from_limbs(out, nl)[0]


def prove_f(field, p, inp, threads=1, mode=0, cap=1 << 25):
    """BfvEncrypt::prove of the C++ oracle over `field` ("goldilocks" | "bn254"); inp holds Goldilocks-form tables."""
    buf = (C.c_uint8 * cap)()
    ln = C.c_size_t(0)
    tm = (C.c_double * 2)()
    err = _err()
    rc = _sym(field, "prove_mode")(C.byref(p.struct), C.byref(inp.struct), threads, mode, buf, C.c_size_t(cap), C.byref(ln), tm, err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return bytes(buf[:ln.value]), (tm[0], tm[1])


def verify_f(field, p, inp, proof, threads=1, mode=0):
    err = _err()
    rc = _sym(field, "verify_mode")(C.byref(p.struct), C.byref(inp.struct), threads, mode, proof, C.c_size_t(len(proof)), err, C.c_size_t(512))
    return rc == 0, err.value.decode()


def lasso_prove_f(field, p, lasso_in_ints, threads=1, mode=0, chain_skip=0, cap=1 << 25):
    """Lasso node of the C++ oracle over `field` on a table of Python integers / numpy u64 (2^nu entries)."""
    fl, el = limbs_of(field)
    nu = int(len(lasso_in_ints)).bit_length() - 1
    lin = to_limbs(lasso_in_ints, fl) if fl > 1 else np.ascontiguousarray(lasso_in_ints, dtype=np.uint64)
    buf = (C.c_uint8 * cap)()
    ln = C.c_size_t(0)
    claim = np.zeros((nu + 1) * el, dtype=np.uint64)
    err = _err()
    rc = _sym(field, "lasso_prove_mode")(C.byref(p.struct), ptr(lin), threads, mode, C.c_size_t(chain_skip), buf, C.c_size_t(cap), C.byref(ln), ptr(claim),
                                        err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return bytes(buf[:ln.value]), from_limbs(claim, el) if field == "bn254" else claim


def lasso_verify_f(field, p, proof, mode=0):
    err = _err()
    rc = _sym(field, "lasso_verify_mode")(C.byref(p.struct), mode, proof, C.c_size_t(len(proof)), err, C.c_size_t(512))
    return rc == 0, err.value.decode()


def grand_product_f(field, tabs, chain_skip=0, threads=1, cap=1 << 24):
    """prove_grand_product of the C++ oracle on nb base-field tables (lists of Python ints); -> (proof bytes, claims, point)."""
    fl, el = limbs_of(field)
    nb, ln_ = len(tabs), len(tabs[0])
    nv = ln_.bit_length() - 1
    arrs = [to_limbs(t, fl) for t in tabs]
    ptrs = (u64p * nb)(*[ptr(a) for a in arrs])
    buf = (C.c_uint8 * cap)()
    ln = C.c_size_t(0)
    claims = np.zeros(nb * el, dtype=np.uint64)
    point = np.zeros(nv * el, dtype=np.uint64)
    err = _err()
    rc = _sym(field, "grand_product")(C.c_size_t(nb), C.c_size_t(ln_), ptrs, C.c_size_t(chain_skip), threads, buf, C.c_size_t(cap), C.byref(ln),
                                     ptr(claims), ptr(point), err, C.c_size_t(512))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return bytes(buf[:ln.value]), from_limbs(claims, el), from_limbs(point, el)


def gl_form(v):
    """A small signed integer given as an Fr element (negatives as r - |z|) -> its Goldilocks residue (p - |z|)."""
    v = int(v)
    return v if v <= R_BN // 2 else P - (R_BN - v)


def bn254_fixture_inputs(n=1024, k=1, bits=27):
    """The reference's bn254 fixture laid out by get_inputs, as Goldilocks-form tables (what the C surface takes)."""
    w = json.load(open(os.path.join(GOLDEN, f"bn254_sk_enc_{n}_{k}x{bits}_65537.json")))
    conv = {}
    for f in ("s", "e", "k1"):
        conv[f] = [gl_form(x) for x in w[f]]
    for f in ("ais", "r1is", "r2is", "ct0is"):
        conv[f] = [[gl_form(x) for x in row] for row in w[f]]
    return Inputs(layout_inputs(n, k, conv))


def elems_be(proof, nbytes):
    return [int.from_bytes(proof[i:i + nbytes], "big") for i in range(0, len(proof), nbytes)]
