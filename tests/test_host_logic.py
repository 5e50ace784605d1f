"""CPU-only tests of the product's host logic and of the C-ABI surface (no GPU compute):
library loads and exports every symbol include/hg.h declares, Fiat-Shamir chain KATs, JSON loader vs a
Python restatement of get_inputs, Lasso preprocessing map, circuit wiring (sum == ct0is), synthetic witness."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orclib
from hglib import hg, ROOT


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hg.h")).read()
    declared = set(re.findall(r"\b(hg_[a-z_0-9]+)\s*\(", hdr))
    lib = C.CDLL(os.path.join(ROOT, "hyper-greco_amd", "libhypergreco.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/hg.h but not exported"
    assert declared == set(hg.EXPORTS)


def test_no_device_fails_loudly():
    if hg.lib().hg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hg.HgError, match="no HIP device"):
        hg.Context(0)


def test_product_never_links_the_oracle():
    # the product sources may not include / link / call anything under oracle/
    src = os.path.join(ROOT, "hyper-greco_amd")
    for dp, _, files in os.walk(src):
        for f in files:
            if f.endswith((".hip", ".cpp", ".hpp", ".inc", ".py", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle/" not in txt.replace("lives in oracle/", "") and "liboracle" not in txt and "orclib" not in txt, f


def test_challenge_chain_kats():
    c = hg.challenges(4)
    assert [int(x) for x in c] == [15017384644633299356, 6854594310142832579, 9149254073876997563, 1396060396769822097]
    # grows consistently
    big = hg.challenges(20000)
    assert (big[:4] == c).all()
    ref = np.zeros(64, dtype=np.uint64)
    orclib.lib().orc_challenge_chain(C.c_size_t(64), orclib.ptr(ref))
    assert (big[:64] == ref).all()


def test_builtin_params_match_reference_constants():
    for key, c in [((1024, 1), orclib.constants(1024, 1)), ((4096, 2), orclib.constants(4096, 2)), ((32768, 16), orclib.constants(32768, 16)),
                   ((2048, 1), orclib.constants(2048, 1)), ((8192, 4), orclib.constants(8192, 4)), ((16384, 8), orclib.constants(16384, 8))]:
        p = hg.params_builtin(*key)
        assert (p.n, p.k, p.s_bound, p.e_bound, p.k1_bound) == (c["n"], c["k"], c["s_bound"], c["e_bound"], c["k1_bound"])
        for f in ("r1_bounds", "r2_bounds", "qis", "k0is"):
            assert list(getattr(p, f))[:p.k] == c[f]
    with pytest.raises(hg.HgError):
        hg.params_builtin(1000, 3)


@pytest.mark.parametrize("n,k,bits", [(1024, 1, 27), (4096, 2, 55)])
def test_json_loader_matches_get_inputs(n, k, bits):
    p = hg.params_builtin(n, k)
    w = hg.Witness.from_json(p, os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    a = w.arrays()
    ref = orclib.fixture_inputs(n, k, bits)
    for f in a:
        assert (a[f] == ref.d[f]).all(), f


def test_json_loader_rejects_malformed(tmp_path):
    p = hg.params_builtin(1024, 1)
    bad = tmp_path / "bad.json"
    bad.write_text('{"s": ["1", "x"]}')
    with pytest.raises(hg.HgError):
        hg.Witness.from_json(p, str(bad))
    bad.write_text('{"s": ["18446744069414584321"]}')  # = p, not canonical
    with pytest.raises(hg.HgError):
        hg.Witness.from_json(p, str(bad))
    with pytest.raises(hg.HgError):
        hg.Witness.from_json(p, str(tmp_path / "missing.json"))


@pytest.mark.parametrize("n,k", [(1024, 1), (4096, 2), (32768, 16)])
def test_lasso_preprocessing_matches_oracle_and_survey(n, k):
    pk = hg.BfvEncrypt.new(n, k).setup(None)  # host-only key
    assert pk.lasso_layout() == orclib.lasso_layout(orclib.params(n, k))
    chunks = max(1, k // 2)
    assert pk.num_nodes == 5 * k + 14 + chunks
    assert pk.rows == (k + chunks + 3) * 2 * n
    pk.free()


@pytest.mark.parametrize("n,k,bits", [(1024, 1, 27), (4096, 2, 55)])
def test_circuit_wiring_on_fixtures(n, k, bits):
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(None)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    lasso_in, sum_out = pk.circuit_eval(w)
    assert (sum_out == w.arrays()["ct0is"]).all()
    o_lasso, o_sum, _ = orclib.circuit_eval(orclib.params(n, k), orclib.Inputs(w.arrays()))
    assert (lasso_in == o_lasso).all() and (sum_out == o_sum).all()
    pk.free()


@pytest.mark.parametrize("n,k", [(1024, 1), (4096, 2), (8192, 4)])
def test_synthetic_witness_satisfies_circuit_and_bounds(n, k):
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(None)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    w2 = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    a = w.arrays()
    assert all((a[f] == w2.arrays()[f]).all() for f in a)  # seeded, deterministic
    lasso_in, sum_out = pk.circuit_eval(w)
    assert (sum_out == a["ct0is"]).all()  # the encryption relation holds in the field
    c = orclib.constants(n, k)
    SZ = 2 * n
    chunks = max(1, k // 2)
    bounds = c["r1_bounds"][:k] + [c["r2_bounds"][0]] * chunks + [c["s_bound"], c["e_bound"], c["k1_bound"]]
    for i, b in enumerate(bounds):
        assert int(lasso_in[i * SZ:(i + 1) * SZ].max()) <= 2 * b
    # and the oracle accepts its own proof on it
    p = orclib.params(n, k)
    inp = orclib.Inputs(a)
    if n <= 4096:
        proof, _ = orclib.prove(p, inp, threads=4)
        ok, err = orclib.verify(p, inp, proof)
        assert ok, err
    pk.free()


@pytest.mark.parametrize("n,k,bits", [(1024, 1, 27), (4096, 2, 55)])
def test_product_verifier_accepts_oracle_proofs_and_rejects_tampering(n, k, bits):
    """hg_verify (product, host) vs the oracle prover: two independent implementations of the same protocol."""
    import random
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(None)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    proof, _ = orclib.prove(orclib.params(n, k), orclib.Inputs(w.arrays()), threads=4)
    ok, why = hg.verify(pk, w, proof)
    assert ok, why
    rng = random.Random(n)
    verdicts = []
    for _ in range(8):
        bad = bytearray(proof)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        ok_p, _ = hg.verify(pk, w, bytes(bad))
        ok_o, _ = orclib.verify(orclib.params(n, k), orclib.Inputs(w.arrays()), bytes(bad))
        assert ok_p == ok_o  # both verifiers bind exactly the same sections
        verdicts.append(ok_p)
    assert verdicts.count(False) >= 5
    ok, _ = hg.verify(pk, w, proof[:-8])
    assert not ok
    d = dict(w.arrays())
    d["ct0is"] = d["ct0is"].copy()
    d["ct0is"][3] ^= np.uint64(1)
    ok, _ = hg.verify(pk, hg.Witness.from_arrays(bfv.params, d), proof)
    assert not ok
    pk.free()


# ---- bn254 family: fixture loader and host verifier (hg_witness_from_json_bn254, hg_verify_bn254) -------------------------------
BN_FIXTURE = os.path.join(orclib.GOLDEN, "bn254_sk_enc_1024_1x27_65537.json")


@pytest.mark.parametrize("n,k,bits", [(2048, 1, 52), (4096, 2, 55)])
def test_bn254_fixture_loader_on_the_larger_reference_fixtures(n, k, bits):
    """hg_witness_from_json_bn254 on the other two bn254 witnesses the reference holds (k = 2: two CRT components, one r2is chunk):
    the arrays equal the test-side layout of the same file and the integer witness satisfies sum == ct0is."""
    bfv = hg.BfvEncrypt.new(n, k)
    w = hg.Witness.from_json_bn254(bfv.params, os.path.join(orclib.GOLDEN, f"bn254_sk_enc_{n}_{k}x{bits}_65537.json"))
    a = w.arrays()
    exp = orclib.bn254_fixture_inputs(n, k, bits).d
    for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is"):
        assert (a[f] == exp[f]).all(), f
    pk = bfv.setup(None)
    _, sum_out = pk.circuit_eval(w)
    assert (sum_out == a["ct0is"]).all()
    pk.free()


@pytest.mark.parametrize("n,k,bits", [(2048, 1, 52), (8192, 4, 55)])
def test_goldilocks_fixture_loader_on_the_larger_reference_fixtures(n, k, bits):
    """hg_witness_from_json on the Goldilocks witnesses beyond c1 / c2 (8192 is the reference's only k = 4 witness: two r2is
    chunks, sk_encryption_circuit.rs:149-161): same arrays as the test-side layout, and the host verifier accepts the oracle's proof."""
    bfv = hg.BfvEncrypt.new(n, k)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    a = w.arrays()
    inp = orclib.fixture_inputs(n, k, bits)
    for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is"):
        assert (a[f] == inp.d[f]).all(), f
    pk = bfv.setup(None)
    _, sum_out = pk.circuit_eval(w)
    assert (sum_out == a["ct0is"]).all()
    proof, _ = orclib.prove(orclib.params(n, k), inp, threads=8)
    assert hg.verify(pk, w, proof) == (True, "")
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    assert not hg.verify(pk, w, bytes(bad))[0]
    pk.free()


def test_bn254_fixture_loader_recovers_the_signed_integers(tmp_path):
    """The reference's bn254 witness holds small signed integers as Fr elements; the loader keeps them in the Goldilocks form.
    Lifted back into Fr they must equal the Python oracle's layout of the same file, and the integer witness satisfies the
    circuit relation over Goldilocks too."""
    import json
    G = orclib.bn254_gkr()
    bfv = hg.BfvEncrypt.new(1024, 1)
    w = hg.Witness.from_json_bn254(bfv.params, BN_FIXTURE)
    a = w.arrays()
    inputs, ct0is = G.layout_inputs(1024, 1, json.load(open(BN_FIXTURE)))
    got = [[G.lift_signed(v) for v in a[f]] for f in ("s", "e", "k1", "ais", "r1is", "r2is")]
    assert got == inputs and [G.lift_signed(v) for v in a["ct0is"]] == ct0is
    pk = bfv.setup(None)
    _, sum_out = pk.circuit_eval(w)
    assert (sum_out == a["ct0is"]).all()
    # an element that is not a small signed integer is refused, not wrapped
    raw = json.load(open(BN_FIXTURE))
    raw["s"][0] = str(1 << 200)
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(raw))
    with pytest.raises(hg.HgError):
        hg.Witness.from_json_bn254(bfv.params, str(bad))
    raw["s"][0] = str(G.R)  # not a canonical field element
    bad.write_text(json.dumps(raw))
    with pytest.raises(hg.HgError):
        hg.Witness.from_json_bn254(bfv.params, str(bad))
    pk.free()


def test_bn254_host_verifier_agrees_with_the_oracle():
    """hg_verify_bn254 (host, no device) accepts the Python oracle's proof of the reference's bn254 fixture and rejects what
    the oracle's verifier rejects: a tampered element at several places, a truncated / extended proof, another witness."""
    import json
    G, bn = orclib.bn254_gkr(), orclib.bn254()
    n, k = 1024, 1
    c = orclib.constants(n, k)
    inputs, ct0is = G.layout_inputs(n, k, json.load(open(BN_FIXTURE)))
    prove_fn, verify_fn = orclib.bn254_lasso_fns(orclib.params(n, k))
    chal = bn.challenges(3000, orclib.keccak256)
    proof, _ = G.prove(c, inputs, ct0is, chal, prove_fn)
    enc = lambda els: b"".join(int(v).to_bytes(32, "big") for v in els)
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(None)
    w = hg.Witness.from_json_bn254(bfv.params, BN_FIXTURE)
    ok, why = hg.verify_bn254(pk, w, enc(proof))
    assert ok, why
    for at in (0, 7, len(proof) // 3, len(proof) // 2, len(proof) - 1):
        bad = list(proof)
        bad[at] = (bad[at] + 1) % G.R
        with pytest.raises(ValueError):
            G.verify(c, inputs, ct0is, bad, chal, verify_fn)
        ok, why = hg.verify_bn254(pk, w, enc(bad))
        assert not ok and why
    assert not hg.verify_bn254(pk, w, enc(proof[:-1]))[0]
    assert hg.verify_bn254(pk, w, enc(proof) + b"\0" * 32)[0]   # like the reference, the verifier does not look past the last element it needs
    assert not hg.verify_bn254(pk, w, enc(proof[:5]) + b"\xff" * 32 + enc(proof[6:]))[0]   # non-canonical element
    other = hg.Witness.from_json(bfv.params, os.path.join(orclib.GOLDEN, "sk_enc_1024_1x27_65537.json"))  # a different sample
    assert not hg.verify_bn254(pk, other, enc(proof))[0]
    pk.free()


# ---- Rust shim crate (rust/hg-shim): the FFI block must mirror include/hg.h signature by signature -------------------------
_C2RUST = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "double": "f64", "uint8_t": "u8",
           "char": "c_char", "void": "c_void", "hg_ctx": "HgCtx", "hg_pk": "HgPk", "hg_witness": "HgWitness", "hg_values": "HgValues", "hg_group": "HgGroup",
           "hg_params": "HgParams", "hg_timings": "HgTimings", "hg_kernel_stat": "HgKernelStat"}


def _c_type_to_rust(t):
    """`const uint64_t* const*` -> `*const *const u64`, written independently of scripts/gen_rust_ffi.py."""
    toks = re.findall(r"const|\*|\w+", t)
    base = [x for x in toks if x not in ("const", "*")][0]
    out = _C2RUST[base]
    const_pending = toks[0] == "const"          # `const T ...`: the pointee of the first `*` is const
    i = toks.index(base) + 1
    while i < len(toks):
        if toks[i] == "const" and i + 1 <= len(toks):    # `* const`: the NEXT pointer level points to a const pointer
            i += 1
            continue
        if toks[i] == "*":
            out = ("*const " if const_pending else "*mut ") + out
            const_pending = i + 1 < len(toks) and toks[i + 1] == "const"
        i += 1
    return out


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "hg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", hdr, flags=re.S)
    fns = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(hg_\w+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[\w*\]$", a)
                if arr:
                    a = a[:arr.start()] .strip()
                ty = re.match(r"(.*?)(\w+)$", a).group(1).strip() + ("*" if arr else "")
                params.append(_c_type_to_rust(ty))
        fns[name] = (params, None if ret == "void" else _c_type_to_rust(ret))
    return fns


def test_rust_shim_ffi_block_mirrors_the_c_header():
    rs = open(os.path.join(ROOT, "rust", "hg-shim", "src", "ffi.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    rust = {}
    for m in re.finditer(r"pub fn (hg_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        rust[m.group(1)] = (args, m.group(3).strip() if m.group(3) else None)
    c = _header_functions()
    assert set(c) == set(rust) == set(hg.EXPORTS), set(c) ^ set(rust)
    for name in sorted(c):
        assert c[name] == rust[name], (name, c[name], rust[name])
    # struct layouts: field order and types of the three by-value structs
    for cname, rname, fields in (("hg_params", "HgParams", ["n: u32", "k: u32", "s_bound: u64", "e_bound: u64", "k1_bound: u64",
                                                            "r1_bounds: [u64; HG_MAX_K]", "r2_bounds: [u64; HG_MAX_K]", "qis: [u64; HG_MAX_K]", "k0is: [u64; HG_MAX_K]"]),
                                 ("hg_timings", "HgTimings", [f + ": f64" for f in ("witness_ms", "upload_ms", "prove_ms", "gpu_ms", "total_ms", "enqueue_ms", "sync_ms", "replay_ms")])):
        body = re.search(r"pub struct %s \{(.*?)\}" % rname, rs, flags=re.S).group(1)
        assert [f.strip()[4:].rstrip(",") for f in body.strip().splitlines()] == fields, rname
    # every function the hand-written shim calls exists in the FFI block, and the generator agrees with the committed file
    used = set()
    for f in ("bfv.rs", "node.rs", "lib.rs"):
        used |= set(re.findall(r"\b(hg_[a-z_0-9]+)\(", open(os.path.join(ROOT, "rust", "hg-shim", "src", f)).read()))
    assert used and used <= set(rust), used - set(rust)
    import subprocess, sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"]).returncode == 0
    # the crate carries the same git dependencies as the reference workspace (Cargo.toml:10,17,28,63-68)
    cargo = open(os.path.join(ROOT, "rust", "hg-shim", "Cargo.toml")).read()
    for dep in ("github.com/han0110/gkr", "github.com/nulltea/gkr-lasso", "github.com/han0110/plonkish", "github.com/nulltea/goldilocks", "halo2curves"):
        assert dep in cargo, dep


def test_witness_json_writer_inverts_get_inputs():
    """scripts/witness_to_json.py (witness -> the reference's BfvSkEncryptArgs JSON) composed with the loader is the identity, and
    it reproduces the reference's own fixture from its laid-out tables."""
    import json, sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import witness_to_json
    for n, k, bits in ((1024, 1, 27), (4096, 2, 55)):
        ref = json.load(open(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json")))
        d = orclib.layout_inputs(n, k, ref)
        back = witness_to_json.arrays_to_args(n, k, d)
        for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is"):
            assert back[f] == ref[f], f
    p = hg.params_builtin(2048, 1)
    w = hg.Witness.synthetic(p, 5)
    args = witness_to_json.arrays_to_args(2048, 1, w.arrays())
    d2 = orclib.layout_inputs(2048, 1, args)
    for f, a in w.arrays().items():
        assert (d2[f] == a).all(), f
    from reference_baseline import parse_span_ms
    assert parse_span_ms("INFO     GKR prove [ 1.88s | 37.12% / 99.31% ]") == 1880.0
    assert parse_span_ms("  GKR prove [ 103ms | 3% ]") == 103.0 and parse_span_ms("nothing") is None


def test_params_derive_reproduces_every_shipped_constant_set():
    """hg_params_derive = the constants emitter of scripts/circuit_sk.py (:80, :249, :296-297, :334-337, :422-439), including the
    script's float rounding of (q - 1) / 2: from (n, qis, t) alone it must reproduce all six constants/*.rs files."""
    for key in ("1024_1", "2048_1", "4096_2", "8192_4", "16384_8", "32768_16"):
        c = orclib.constants(*[int(x) for x in key.split("_")])
        p = hg.params_derive(c["n"], c["k"], c["qis"], 65537)
        b = hg.params_builtin(c["n"], c["k"])
        assert (p.n, p.k, p.s_bound, p.e_bound, p.k1_bound) == (b.n, b.k, b.s_bound, b.e_bound, b.k1_bound), key
        for f in ("r1_bounds", "r2_bounds", "qis", "k0is"):
            assert list(getattr(p, f)) == list(getattr(b, f)), (key, f)
    # a set the reference does not ship: n = 512 with one 27-bit modulus still gives a provable parameter set
    p = hg.params_derive(512, 1, [orclib.constants(1024, 1)["qis"][0]], 65537)
    assert p.n == 512 and p.r1_bounds[0] > 0 and p.r2_bounds[0] == (orclib.constants(1024, 1)["qis"][0] - 1) // 2
    with pytest.raises(hg.HgError):
        hg.params_derive(1000, 1, [12289], 65537)
    with pytest.raises(hg.HgError):
        hg.params_derive(1024, 3, [12289, 12289, 12289], 65537)


def test_witness_for_other_parameters_is_refused():
    """A witness handle built for one parameter set used with the prover key of another: every entry point that
    indexes the witness with the key's sizes refuses instead of reading out of bounds."""
    small, big = hg.BfvEncrypt.new(1024, 1), hg.BfvEncrypt.new(4096, 2)
    pk_big = big.setup(None)           # host-only key: enough for the host-side entry points
    w_small = hg.Witness.synthetic(small.params, 3)
    L = hg.lib()
    rc = L.hg_verify(pk_big.h, w_small.h, b"\0" * 64, 64)
    assert rc == -1 and b"witness was built for n=1024 k=1" in L.hg_last_error()
    rc = L.hg_verify_bn254(pk_big.h, w_small.h, b"\0" * 64, 64)
    assert rc == -1 and b"witness was built for" in L.hg_last_error()
    with pytest.raises(hg.HgError, match="witness was built for"):
        pk_big.circuit_eval(w_small)
    pk_big.free()


def test_reference_baseline_leaves_the_reference_fixture_as_it_was(tmp_path, monkeypatch):
    """scripts/reference_baseline.py writes the synthetic witness where the reference's test reads it - the shipped fixture at that
    path must be back afterwards, byte for byte, whether cargo succeeds, fails or cannot be started."""
    import stat
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import reference_baseline
    repo = tmp_path / "hyper-greco"
    data = repo / "bfv-gkr" / "src" / "data" / "goldilocks"
    data.mkdir(parents=True)
    fixture = data / "sk_enc_1024_1x27_65537.json"
    original = b'{"shipped": "fixture"}\n'
    fixture.write_bytes(original)
    bindir = tmp_path / "bin"
    bindir.mkdir()
    cargo = bindir / "cargo"
    seen = tmp_path / "seen.txt"
    cargo.write_text("#!/bin/sh\nwc -c < %s >> %s\necho 'INFO GKR prove [ 1.50s | 37.12%% / 99.31%% ]'\nexit ${FAKE_CARGO_RC:-0}\n" % (fixture, seen))
    cargo.chmod(cargo.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("HYPER_GRECO", str(repo))
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    r = reference_baseline.measure(1024, 1, 5, runs=1)
    assert r and r["kind"] == "reference" and abs(r["value"] - 1500.0) < 1e-6
    assert fixture.read_bytes() == original and not (data / "sk_enc_1024_1x27_65537.json.hg-backup").exists()
    assert int(seen.read_text().split()[0]) > 1000      # cargo saw the synthetic witness, not the shipped file
    monkeypatch.setenv("FAKE_CARGO_RC", "1")
    assert reference_baseline.measure(1024, 1, 5, runs=1) is None
    assert fixture.read_bytes() == original
    # a configuration whose fixture is a missing blob: nothing is left behind
    assert reference_baseline.measure(2048, 1, 5, runs=1) is None
    assert not (data / "sk_enc_2048_1x52_65537.json").exists()


def test_setup_finds_the_eq_factored_nodes():
    """hg_setup's classification of the Vanilla nodes (hg_pk::NodeDev::EqForm, host-only key): the wirings of BfvEncryptBlock::configure
    [REF sk_encryption_circuit.rs:97-285] that relay aligned blocks - es, k1kis (k blocks of one input each), r1iqis (times q_i), the
    chunk nodes (ONE block, a window of r2is), lasso_inputs_batched (with its additive bounds), s_eval_copy, sai_par and the final sum -
    are eq-factored; the k sai_eval nodes (mul gates) and r2i_cyclo (two relays per position plus a zero gate) are not."""
    for n, k in ((1024, 1), (4096, 2), (32768, 16)):
        bfv = hg.BfvEncrypt.new(n, k)
        pk = bfv.setup(None)
        rows = [pk.node_eq_form(i) for i in range(pk.num_nodes)]
        van = [r for r in rows if r["kind"] == "vanilla"]
        L = n.bit_length()             # log2_size = N_LOG2 + 1
        chunks = max(1, (n * k) // (2 * n))   # r2is has n k entries, a chunk node relays 2n of them
        assert len(van) == 7 + chunks + k + 1, (n, k, len(van))   # es k1kis r1iqis lasso_in s_eval_copy sai_par sum + chunks + sai_eval + cyclo
        eq = [r for r in van if r["eq_form"]]
        assert len(eq) == 7 + chunks, (n, k, [(r["in_log2"], r["terms"]) for r in eq])
        assert len([r for r in van if not r["eq_form"]]) == k + 1
        lasso_in = rows[pk.lasso_in_id]
        assert lasso_in["eq_form"] and lasso_in["block_log2"] == L and lasso_in["terms"] == k + chunks + 3 and lasso_in["window"] == 0
        total = rows[pk.sum_id]
        assert total["eq_form"] and total["terms"] == 5 and total["block_log2"] == total["in_log2"] == L + (k.bit_length() - 1)
        wins = sorted(r["window"] for r in eq if r["in_log2"] > r["block_log2"])   # the chunk nodes: window c of the r2is table
        assert wins == list(range(chunks)) or chunks == 1, (n, k, wins)
        pk.free()


def test_bench_issue_model_reads_the_committed_rate_table():
    """bench.py prices a kernel's VALU instructions per instruction class (roofline.issue_frac): the two rates come from the committed output
    of scripts/ub/ratebench.hip (profiles/r06_ratebench.txt). The plain VOP1 / VOP2 integer forms must come out at about twice the rate of
    everything else, and a kernel's issue time must lie between the all-class-A and the all-class-B price of its instruction count."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hg_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    im = bench.IssueModel()
    assert im.rate_a and im.rate_b, im.note
    assert 0.9e9 < im.rate_a < 1.2e9 and 0.45e9 < im.rate_b < 0.65e9 and 1.6 < im.rate_a / im.rate_b < 2.2, (im.rate_a, im.rate_b)
    if im.ok():   # (the class mix is refused when it was taken on other sources: then only the rates are checked)
        sym = bench.CLASS_SYMBOL[bench.ROOFLINE_CLASS]
        fa = im.share_a(sym)
        assert fa is not None and 0.05 < fa < 0.5, fa
        n = 25.5e6
        t = im.seconds(n, sym)
        assert n / im.rate_a / bench.NSIMD < t < n / im.rate_b / bench.NSIMD
