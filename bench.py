#!/usr/bin/env python3
"""Benchmark of the Huffman block-codec hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload const41|zipf255|uniform256|uniform255|logtext]

One "step" = one encode + one decode of the rank's shard (default 1 GiB, 64 KiB blocks) with
the input already resident in HBM.  N > 1 is launched by torch.distributed.run with one rank
per GPU; blocks are independent, so every rank owns a contiguous range of blocks of the one
logical input (weak scaling: bytes per GPU are fixed).  The only exchange between ranks is the
all-gather of the per-rank compressed sizes that places each rank's stream in the global
stream (RCCL, 8 bytes per rank per step).

Prints ONE JSON line on rank 0.  `value` = uncompressed bytes of the whole job per second of
(encode + decode), in GiB/s.  `roofline` is the dominant kernel's algorithmic HBM bytes over
its HIP-event-measured duration inside the timed region; `cpu_baseline` is the unmodified
reference library (oracle/_ref) timed on this box's host CPU on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_EVERY = 4              # steps of the timed region that carry per-kernel HIP events: 0, 4, 8, ...

WORKLOADS = {
    # name -> (BASELINE.json config it is, description)
    "const41": "configs[1]: 1 GiB repeating 0x41, blocksize=64KiB (degenerate one-symbol tree)",
    "zipf255": "configs[2]: 1 GiB Zipf-distributed bytes (zipf255 seed 3), blocksize=64KiB",
    "uniform256": "configs[3] per-GPU share: uniform-random bytes (uniform256 seed 1), blocksize=64KiB, relaxed-tree decode",
    "uniform255": "config 4b: uniform over 255 symbols (seed 2), blocksize=64KiB",
    "logtext": "configs[4] per-GPU share: synthetic log text (16 MiB generator tile repeated), blocksize=1MiB",
}


def cpu_baseline(workload: str, blocksize: int) -> dict:
    """Unmodified reference (oracle/_ref/libhuffman_ref.so) on ONE host core, bounded sample."""
    import numpy as np
    from libhuffman_amd import datagen
    from oracle.oracle import Oracle, Reference
    sample_bytes = {"const41": 256 << 20, "zipf255": 64 << 20, "uniform255": 64 << 20,
                    "uniform256": 64 << 20, "logtext": 64 << 20}[workload]
    data = datagen.GENERATORS[workload](sample_bytes)
    kind = "reference"
    try:
        if not Reference.available():
            raise FileNotFoundError
        ref = Reference()
        t0 = time.perf_counter()
        enc = ref.encode(data, blocksize)
        t1 = time.perf_counter()
        if workload == "uniform256":
            # the reference cannot decode k = 256 blocks (src/decoder.c:237-239): its decode leg is
            # taken from the restatement in relaxed mode and labelled as such
            raise RuntimeError("reference cannot decode k=256")
        err, back = ref.decode(enc, raw_hint=sample_bytes + 64)
        t2 = time.perf_counter()
        assert err == 0 and back.size == sample_bytes
        t_enc, t_dec = t1 - t0, t2 - t1
    except Exception:
        kind = "port"
        ora = Oracle()
        t0 = time.perf_counter()
        enc = ora.encode(data, blocksize)
        t1 = time.perf_counter()
        err, back, _ = ora.decode(enc, sample_bytes, 1025)
        t2 = time.perf_counter()
        assert err == 0 and back.size == sample_bytes
        t_enc, t_dec = t1 - t0, t2 - t1
    # the same library on every host core: blocks are independent, so the sample is cut into one
    # block-aligned slice per core and the slices are encoded / decoded concurrently (ctypes
    # releases the GIL).  Reported beside the 1-thread figure, which stays the baseline `value`.
    all_cores = None
    if kind == "reference":
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncpu = os.cpu_count() or 1
            per = max(blocksize, (sample_bytes // ncpu) // blocksize * blocksize)
            parts = [data[i:i + per] for i in range(0, sample_bytes, per)]
            with ThreadPoolExecutor(ncpu) as ex:
                t0 = time.perf_counter()
                encs = list(ex.map(lambda part: ref.encode(part, blocksize), parts))
                t1 = time.perf_counter()
                backs = list(ex.map(lambda e: ref.decode(e, raw_hint=per + 64), encs))
                t2 = time.perf_counter()
            assert all(err == 0 for err, _ in backs) and sum(b.size for _, b in backs) == sample_bytes
            all_cores = {"value": round(sample_bytes / GIB / (t2 - t0), 5), "unit": "GiB/s", "cores": ncpu,
                         "threads": min(ncpu, len(parts))}
        except Exception as e:                                  # never fail the bench over the extra figure
            all_cores = {"error": repr(e)}
    return {"value": round(sample_bytes / GIB / (t_enc + t_dec), 5), "unit": "GiB/s", "cores": 1,
            "kind": kind, "all_cores": all_cores,
            "sample": f"{sample_bytes >> 20} MiB of {workload}, {blocksize >> 10} KiB blocks, "
                      f"encode {t_enc:.2f}s + decode {t_dec:.2f}s, memstreams, 1 thread",
            "encode_GiBps": round(sample_bytes / GIB / t_enc, 5),
            "decode_GiBps": round(sample_bytes / GIB / t_dec, 5)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="const41", choices=sorted(WORKLOADS))
    ap.add_argument("--bytes-per-gpu", type=int, default=1 << 30)
    ap.add_argument("--blocksize", type=int, default=None, help="default 64 KiB (1 MiB for logtext)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the codec has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # the latter: 1-rank test of the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from libhuffman_amd.codec import GpuCodec
    from libhuffman_amd.sharding import shard_range

    codec = GpuCodec(local_rank)
    bs = args.blocksize or ((1 << 20) if args.workload == "logtext" else 65536)
    n_total = args.bytes_per_gpu * world
    lo, hi = shard_range(n_total, bs, rank, world)          # contiguous block range of this rank
    n = hi - lo
    nb = codec.block_count(n, bs)
    relaxed = args.workload == "uniform256"

    dev = torch.device("cuda", local_rank)
    if args.workload == "logtext":
        # the text generator runs on the host: one 16 MiB tile, repeated on the device
        from libhuffman_amd import datagen
        tile = torch.from_numpy(datagen.logtext(16 << 20)).to(dev)
        data = tile.repeat((n + tile.numel() - 1) // tile.numel())[:n].contiguous()
    else:
        data = torch.empty(n, dtype=torch.uint8, device=dev)
        codec.fill(data, args.workload, first=lo)
    out = torch.empty(codec.encode_bound(n, bs), dtype=torch.uint8, device=dev)
    offs = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)

    pending = []

    def step():
        codec.encode(data, bs, out=out, offsets=offs, sync=False)
        if use_dist:
            # the one real exchange: every rank learns where its stream starts in the job's stream.
            # Nothing on this rank's decode depends on it, so it runs on RCCL's stream beside the
            # decode (it waits for the encode by itself) and is only waited for at the end.
            if os.environ.get("BENCH_SYNC_GATHER") == "1":          # (A/B switch: the collective in line)
                dist.all_gather_into_tensor(sizes, offs[nb:nb + 1])
            else:
                pending.append(dist.all_gather_into_tensor(sizes, offs[nb:nb + 1], async_op=True))
        codec.decode(out, out.numel(), offs, nb, back, relaxed=relaxed, sync=False)

    def drain():
        while pending:
            pending.pop().wait()

    for _ in range(args.warmup):
        step()
    drain()
    raw = codec.decode_result() if args.warmup else None
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    # HIP events around every kernel cost ~5 us each (~6 % of a config-2 step), so inside the timed
    # region every PROFILE_EVERY-th step carries them; the per-kernel averages are over those steps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        codec.set_profiling(k % PROFILE_EVERY == 0, resume=k > 0)
        step()
    drain()
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    raw = codec.decode_result()
    assert raw == n, f"decode produced {raw} of {n} bytes"
    enc_prof, enc_calls = codec.profile("encode")
    dec_prof, dec_calls = codec.profile("decode")
    codec.set_profiling(False)

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    comp_len = int(offs[nb].item())
    bit_exact = None
    if not args.no_verify:
        bit_exact = bool(torch.equal(back, data))      # full-size round trip on every rank
        if use_dist:
            ok = torch.tensor([1 if bit_exact else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            bit_exact = bool(ok.item())

    if rank == 0:
        K = args.steps
        value = n_total * K / GIB / elapsed
        # per-launch algorithmic bytes: encode reads N writes C, decode reads C writes N
        alg = {"pack": n + comp_len, "decode": comp_len + n, "hist256": n, "tree": nb * (1024 + 2048 + 2064),
               "scan_sizes": nb * 24, "prepare_scan": nb * (16 + 10 + 28)}
        if args.workload == "const41":
            alg["pack"] = comp_len          # one-symbol blocks: the input is not read again, the payload is zeros
        # blocks below 4 MiB take the fused histogram+tree kernel: its time is reported once
        fused = bs < (1 << 22)
        if fused:
            enc_prof = dict(enc_prof)
            # (the "tree" and "scan_sizes" stages are empty event gaps: that work runs inside the fused kernel)
            enc_prof["hist_tree"] = enc_prof.pop("hist256") + enc_prof.pop("tree") + enc_prof.pop("scan_sizes")
            alg["hist_tree"] = n
        kernels = {}
        for name, ms in list(enc_prof.items()) + list(dec_prof.items()):
            calls = enc_calls if name in enc_prof else dec_calls
            avg_ms = ms / max(calls, 1)
            kernels[name] = {"avg_ms": round(avg_ms, 4),
                             "alg_GBps": round(alg[name] / 1e9 / (avg_ms / 1e3), 1) if avg_ms > 0 else None}
        dom = max((k for k in ("pack", "decode", "hist256", "tree", "hist_tree") if k in kernels),
                  key=lambda k: kernels[k]["avg_ms"])
        achieved = alg[dom] / 1e9 / (kernels[dom]["avg_ms"] / 1e3)
        # HBM bytes of that kernel from the TCC counters (separate rocprofv3 --pmc passes of this same
        # command, tools/gpu_traffic.sh -> profiles/traffic.json); null when not collected for the workload
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                t = json.load(f)["workloads"][args.workload][dom]
            if n == (1 << 30) and bs == 65536:
                traffic = round(t["hbm"])
        except Exception:
            traffic = None
        pipeline_bytes = 2 * (n + comp_len)
        gpu_ms = ev0.elapsed_time(ev1) / K
        # encode-only / decode-only: this rank's bytes over the kernels of each half
        enc_ms = sum(v["avg_ms"] for k, v in kernels.items() if k in enc_prof)
        dec_ms = sum(v["avg_ms"] for k, v in kernels.items() if k in dec_prof)
        result = {
            "metric": "encode+decode GiB/s (uncompressed) on %s blocks" % ("64KiB" if bs == 65536 else "%dKiB" % (bs >> 10)),
            "value": round(value, 3),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload], "generator": args.workload,
                       "bytes_per_gpu": n, "blocksize": bs, "blocks_per_gpu": nb,
                       "compressed_bytes_per_gpu": comp_len, "ratio": round(comp_len / n, 5),
                       "parallelism": f"block-sharded x{world}", "bit_exact_roundtrip": bit_exact},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "alg_bytes_per_launch": alg[dom],
                         "pipeline_frac": round(pipeline_bytes / 1e9 / (gpu_ms / 1e3) / HBM_PEAK_GBS, 4)},
            "kernels": kernels,
            "encode_only_GiBps_per_gpu": round(n / GIB / (enc_ms / 1e3), 1) if enc_ms > 0 else None,
            "decode_only_GiBps_per_gpu": round(n / GIB / (dec_ms / 1e3), 1) if dec_ms > 0 else None,
            "gpu_ms_per_step_rank0": round(gpu_ms, 4),
            "profiled_steps": max(enc_calls, dec_calls),
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload, bs)

    # RCCL writes its version banner through C stdio, which a pipe only sees when a process exits:
    # every rank pushes its buffer out before the last barrier, so that rank 0's JSON line is the
    # last line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
