"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 job (python -m torch.distributed.run ... dist_worker.py <mode>).

mode "combine" (CPU only): the caller-side exchange of a sharded proof without the device part - every rank builds the partial
    result buffer it would hold (seeded canonical lanes, zeros where it "owns nothing", lanes at p-1 on EVERY rank), the buffers
    are all-gathered over gloo and summed by hg_shard_combine_host; every rank checks the result against Python integers.
mode "prove" (needs a GPU; both ranks on device 0): the real two-process sharded prove - hg_prove_shard_begin on each rank's own
    context, gloo all-gather of the partial buffers, hg_prove_shard_combine, hg_prove_shard_finish - at n=4096 k=2; every rank
    must produce the CPU oracle's proof bytes. Five proofs in a row (walk, walk, launch-graph capture, replays) with two witnesses.
mode "prove_c3" (needs a GPU; both ranks on device 0): the same at BASELINE config 4's size, n=32768 k=16, with PER-RANK tables: every
    process evaluates only the cone of its own share (hg_witness_gen_shard: the other rank's per-modulus chains are never computed or
    resident), proves its share, exchanges the partial buffers over gloo; the second witness goes in through hg_witness_gen_into.
    Every rank's proof must be the CPU oracle's.
mode "prove_seq" (needs a GPU; both ranks on device 0): the round-by-round prover of protocol mode 3 (absorbing transcript +
    extension-field memory checking) on two processes, ONE all-reduce per sum-check round (hg_prove_resident_mode_sharded with an
    external group: the six words of a round's partial sums are all-gathered over gloo and added mod p); n=1024 k=1 and n=4096 k=2,
    every rank's proof must be the CPU oracle's proof of that mode.
mode "prove_seq_own" (needs a GPU; both ranks on device 0): the same prover at n=32768 k=16 WITHOUT replicating the witness: per-rank tables
    (hg_witness_gen_shard) and node ownership - a Vanilla / FFT node's reduction runs on its owner alone, the other process joins the same
    all-reduces with zeros; every rank's proof must be the CPU oracle's mode-3 proof, with less than the full set of tables resident.
mode "prove_rccl" (needs TWO GPUs: rank r on device r): hg_prove_sharded - the library's own ncclAllReduce of the result buffer over
    xGMI behind each rank's share, per-rank tables (hg_witness_gen_shard) - at n=4096 k=2 and n=32768 k=16; gloo only carries the
    128-byte RCCL id. Every rank's proof must be the CPU oracle's. (RCCL refuses two ranks on one device, so this cannot be faked.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

P = 0xFFFFFFFF00000001


def all_gather_u64(part, world):
    t = torch.from_numpy(part.view(np.int64).copy())
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    return torch.stack(bufs).numpy().view(np.uint64)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    hg = entry.load_package()
    if mode == "combine":
        n = 4099
        rng = np.random.default_rng(1234 + rank)
        part = (rng.integers(0, P, size=n, dtype=np.uint64)).astype(np.uint64)
        part[rank::3] = 0            # lanes this rank does not own
        part[:16] = P - 1            # the largest canonical value on every rank
        part[16:32] = 0xFFFFFFFF     # and the limb boundaries
        part[32:48] = 0x100000000
        gathered = all_gather_u64(part, world)
        assert np.array_equal(gathered[rank], part)
        got = hg.shard_combine_host(gathered)
        want = np.array([sum(int(gathered[r][i]) for r in range(world)) % P for i in range(n)], dtype=np.uint64)
        assert np.array_equal(got, want)
        bad = gathered.copy()
        bad[0][5] = P                # non-canonical lane: refused
        try:
            hg.shard_combine_host(bad)
            raise SystemExit("non-canonical lane accepted")
        except hg.HgError:
            pass
        dist.barrier()
        print("rank %d COMBINE OK" % rank, flush=True)
    elif mode == "prove":
        import orclib
        n, k = 4096, 2
        ctx = hg.Context(0)
        bfv = hg.BfvEncrypt.new(n, k)
        pk = bfv.setup(ctx)
        ws = [hg.Witness.synthetic(bfv.params, 0x51 + j) for j in range(2)]
        refs = [orclib.prove(orclib.params(n, k), orclib.Inputs(w.arrays()), threads=4)[0] for w in ws]
        vals = hg.witness_gen(ctx, pk, ws[0])
        out = hg.ProofBuffer()
        for it in range(5):          # walk, walk, capture, then a NEW witness through the same graph, twice
            j = 0 if it < 3 else 1
            if it == 3:
                hg.witness_gen_into(ctx, pk, ws[1], vals)
            part = hg.prove_shard_begin(ctx, pk, vals, rank, world)
            gathered = all_gather_u64(np.array(part, copy=True), world)
            hg.prove_shard_combine(ctx, gathered, world)
            got = hg.prove_shard_finish(ctx, out).bytes()
            assert got == refs[j], "rank %d, proof %d: sharded two-process proof differs from the CPU oracle" % (rank, it)
        dist.barrier()
        print("rank %d PROVE OK" % rank, flush=True)
        vals.free(); pk.free(); ctx.close()
    elif mode == "prove_c3":
        import orclib
        n, k = 32768, 16
        ctx = hg.Context(0)
        bfv = hg.BfvEncrypt.new(n, k)
        pk = bfv.setup(ctx)
        ws = [hg.Witness.synthetic(bfv.params, 0x61 + j) for j in range(2)]
        threads = max(2, min(16, (os.cpu_count() or 4) // 2))
        refs = [orclib.prove(orclib.params(n, k), orclib.Inputs(w.arrays()), threads=threads)[0] for w in ws]
        vals = hg.witness_gen_shard(ctx, pk, ws[0], rank, world)
        info = vals.info()
        assert info["resident_bytes"] < info["full_bytes"] and info["resident_tables"] < info["tables"], info
        assert info["resident_bytes"] < info["peak_bytes"] < info["full_bytes"] + info["resident_bytes"], info   # its tables + the cone, not the circuit
        out = hg.ProofBuffer()
        for it in range(4):          # walk, walk, then the second witness (refill in place): capture, replay
            j = 0 if it < 2 else 1
            if it == 2:
                hg.witness_gen_into(ctx, pk, ws[1], vals)
            part = hg.prove_shard_begin(ctx, pk, vals, rank, world)
            gathered = all_gather_u64(np.array(part, copy=True), world)
            hg.prove_shard_combine(ctx, gathered, world)
            got = hg.prove_shard_finish(ctx, out).bytes()
            assert got == refs[j], "rank %d, proof %d: sharded two-process proof (per-rank tables, n=32768) differs from the CPU oracle" % (rank, it)
        dist.barrier()
        print("rank %d PROVE_C3 OK resident %.1f MB peak %.1f MB of %.1f MB" % (rank, info["resident_bytes"] / 1e6, info["peak_bytes"] / 1e6, info["full_bytes"] / 1e6), flush=True)
        vals.free(); pk.free(); ctx.close()
    elif mode == "prove_seq":
        import orclib
        ctx = hg.Context(0)
        for n, k in ((1024, 1), (4096, 2)):
            bfv = hg.BfvEncrypt.new(n, k)
            pk = bfv.setup(ctx)
            w = hg.Witness.synthetic(bfv.params, 0x71 + n)
            ref = orclib.prove_f("goldilocks", orclib.params(n, k), orclib.Inputs(w.arrays()), threads=4, mode=3)[0]
            vals = hg.witness_gen(ctx, pk, w)
            calls = [0]

            def reduce(words):   # the round's all-reduce: gather every rank's partial sums, add them as field elements, in place
                g = all_gather_u64(np.array(words, copy=True), world)
                acc = [0] * len(words)
                for r in range(world):
                    for i in range(len(words)):
                        acc[i] = (acc[i] + int(g[r][i])) % P
                for i in range(len(words)):
                    words[i] = acc[i]
                calls[0] += 1

            group = hg.Group.external(reduce, world)
            out = hg.ProofBuffer()
            got = hg.prove_resident_mode_sharded(ctx, pk, vals, out, 3, rank, group).bytes()
            assert got == ref, "rank %d: the two-process round-by-round proof (n=%d) differs from the CPU oracle's mode-3 proof" % (rank, n)
            assert calls[0] == int(out.timings()["replay_ms"]) > 100, (calls[0], out.timings())
            vals.free(); pk.free()
        dist.barrier()
        print("rank %d PROVE_SEQ OK (%d all-reduces for the last proof)" % (rank, calls[0]), flush=True)
        ctx.close()
    elif mode == "prove_seq_own":
        import orclib
        ctx = hg.Context(rank if hg.device_count() >= world else 0)   # one GPU per rank where the box has them (the exchange stays on gloo)
        n, k = 32768, 16
        bfv = hg.BfvEncrypt.new(n, k)
        pk = bfv.setup(ctx)
        w = hg.Witness.synthetic(bfv.params, 0x91 + n)
        threads = max(2, min(16, (os.cpu_count() or 4) // world))
        ref = orclib.prove_f("goldilocks", orclib.params(n, k), orclib.Inputs(w.arrays()), threads=threads, mode=3)[0]
        vals = hg.witness_gen_shard(ctx, pk, w, rank, world)      # this rank's share of the node tables only
        info = vals.info()
        assert info["resident_bytes"] < info["full_bytes"] and info["resident_tables"] < info["tables"], info
        calls = [0]

        def reduce(words):   # a round's (or a shared result's) all-reduce: gather, add as field elements, in place - vectorised, the messages reach 144 words
            g = all_gather_u64(np.array(words, copy=True), world)
            acc = np.zeros(len(words), dtype=object)
            for r in range(world):
                acc = (acc + g[r].astype(object)) % P
            for i in range(len(words)):
                words[i] = int(acc[i])
            calls[0] += 1

        group = hg.Group.external(reduce, world)
        out = hg.ProofBuffer()
        got = hg.prove_resident_mode_sharded(ctx, pk, vals, out, 3, rank, group).bytes()
        assert got == ref, "rank %d: the two-process round-by-round proof with node ownership (n=%d k=%d) differs from the CPU oracle's mode-3 proof" % (rank, n, k)
        assert calls[0] == int(out.timings()["replay_ms"]) > 1000, (calls[0], out.timings())
        dist.barrier()
        print("rank %d PROVE_SEQ_OWN OK resident %.1f MB of %.1f MB, %d all-reduces, %.0f ms" % (rank, info["resident_bytes"] / 1e6, info["full_bytes"] / 1e6, calls[0], out.timings()["prove_ms"]), flush=True)
        vals.free(); pk.free(); ctx.close()
    elif mode == "prove_rccl":
        import orclib
        assert hg.device_count() >= world, "prove_rccl needs one GPU per rank"
        ctx = hg.Context(rank)
        uid = [hg.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        hg.comm_init(ctx, uid[0], rank, world)
        assert hg.comm_count(ctx) == world
        for n, k in ((4096, 2), (32768, 16)):
            bfv = hg.BfvEncrypt.new(n, k)
            pk = bfv.setup(ctx)
            ws = [hg.Witness.synthetic(bfv.params, 0x81 + j + n) for j in range(2)]
            threads = max(2, min(16, (os.cpu_count() or 4) // world))
            refs = [orclib.prove(orclib.params(n, k), orclib.Inputs(w.arrays()), threads=threads)[0] for w in ws]
            vals = hg.witness_gen_shard(ctx, pk, ws[0], rank, world)
            out = hg.ProofBuffer()
            for it in range(5):      # walk, walk, capture, then the second witness through the same graph, twice
                j = 0 if it < 3 else 1
                if it == 3:
                    hg.witness_gen_into(ctx, pk, ws[1], vals)
                got = hg.prove_sharded(ctx, pk, vals, out).bytes()
                assert got == refs[j], "rank %d, n=%d, proof %d: the proof sharded over RCCL differs from the CPU oracle" % (rank, n, it)
            vals.free(); pk.free()
        dist.barrier()
        hg.comm_destroy(ctx)
        print("rank %d PROVE_RCCL OK" % rank, flush=True)
        ctx.close()
    else:
        raise SystemExit("unknown mode")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
