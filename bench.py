#!/usr/bin/env python3
"""Benchmark of the Huffman block-codec hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload zipf255|uniform256|const41|uniform255|logtext]

One "step" = one encode + one decode of the rank's shard (default 1 GiB, 64 KiB blocks) with
the input already resident in HBM.  The headline workload is `zipf255` (BASELINE.json configs[2]:
text-like bytes, every block a full 255-symbol tree; the unmodified reference can decode it, so it
is the workload the CPU baseline is timed on); `uniform256` (configs[3], per-GPU share) and
`const41` (configs[1], the degenerate one-symbol tree) run in the same process with the same K/W
and are reported under "secondary" of the same JSON line.

N > 1: one rank per GPU.  Either the driver starts the ranks (torch.distributed.run sets
WORLD_SIZE), or `python bench.py --gpus N` alone starts them itself - as a child process, before
this process touches a GPU - and relays rank 0's line.  Blocks are independent, so every rank owns
a contiguous range of blocks of the one logical input (weak scaling: bytes per GPU are fixed) and
the timed step needs ONE exchange: the all-gather of the per-rank compressed sizes that places each
rank's stream in the job's stream (RCCL, 8 bytes per rank).  Beside that resident-data figure
(`value`), "root_placement" times the form north_star words: the input scattered from rank 0 over
xGMI, encoded, the compressed shards gathered on rank 0, scattered again, decoded, and the output
gathered on rank 0.

Prints ONE JSON line on rank 0.  `value` = uncompressed bytes of the whole job per second of
(encode + decode), in GiB/s.  `roofline` is the dominant kernel's algorithmic HBM bytes over
its HIP-event-measured duration inside the timed region; `cpu_baseline` is the unmodified
reference library (oracle/_ref) timed on this box's host CPU on a bounded sample.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMD_CLOCK_HZ = 2.4e9            # peak engine clock (MI355X_MICROARCH.md); used for roofline.valu_issue_frac only
PROFILE_EVERY = 4              # steps of the timed region that carry per-kernel HIP events: 0, 4, 8, ...

WORKLOADS = {
    # name -> (BASELINE.json config it is, description)
    "zipf255": "configs[2]: 1 GiB Zipf-distributed bytes (zipf255 seed 3), blocksize=64KiB",
    "uniform256": "configs[3] per-GPU share: uniform-random bytes (uniform256 seed 1), blocksize=64KiB, relaxed-tree decode",
    "const41": "configs[1]: 1 GiB repeating 0x41, blocksize=64KiB (degenerate one-symbol tree)",
    "uniform255": "config 4b: uniform over 255 symbols (seed 2), blocksize=64KiB",
    "logtext": "configs[4] per-GPU share: synthetic log text (16 MiB generator tile repeated), blocksize=1MiB",
}
DEFAULT_SECONDARY = {"zipf255": ["uniform256", "const41"]}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(workload: str, blocksize: int) -> dict:
    """Unmodified reference (oracle/_ref/libhuffman_ref.so) on ONE host core, bounded sample."""
    import numpy as np  # noqa: F401
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
    # the same library on every host core: one PROCESS per core (a child interpreter that never touches the GPU starts
    # them), each with its own 8 MiB of the workload, all released together - see cpu_all_cores_child().
    all_cores = None
    if kind == "reference":
        try:
            import signal
            import subprocess
            # (its own process group: a child that overruns is killed with the workers it forked, not left to them)
            pr = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-all-cores-child", workload, str(blocksize)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                                  env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            try:
                stdout, _ = pr.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                pr.communicate()
                raise
            all_cores = json.loads(stdout.strip().splitlines()[-1])
        except Exception as e:                                  # never fail the bench over the extra figure
            all_cores = {"error": repr(e)}
    return {"value": round(sample_bytes / GIB / (t_enc + t_dec), 5), "unit": "GiB/s", "cores": 1,
            "kind": kind, "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "all_cores": all_cores,
            "sample": f"{sample_bytes >> 20} MiB of {workload}, {blocksize >> 10} KiB blocks, "
                      f"encode {t_enc:.2f}s + decode {t_dec:.2f}s, memstreams, 1 thread",
            "encode_GiBps": round(sample_bytes / GIB / t_enc, 5),
            "decode_GiBps": round(sample_bytes / GIB / t_dec, 5)}


def _all_cores_worker(workload, blocksize, per, idx, barrier, q):
    """One core's share of the all-cores baseline: its own `per` bytes of the workload (seeded by the worker's number),
    generated before the barrier; the reference's encode + decode timed behind it."""
    try:
        from libhuffman_amd import datagen
        from oracle.oracle import Reference
        gen = {"zipf255": lambda n: datagen.zipf255(n, 3 + idx), "uniform255": lambda n: datagen.uniform255(n, 2 + idx),
               "uniform256": lambda n: datagen.uniform256(n, 1 + idx)}.get(workload, datagen.GENERATORS[workload])
        data = gen(per)
        ref = Reference()
        ref.encode(data[:blocksize], blocksize)               # (library loaded, pages touched)
        barrier.wait(timeout=120)
        t0 = time.perf_counter()
        enc = ref.encode(data, blocksize)
        t1 = time.perf_counter()
        ok = True
        if workload != "uniform256":                          # (the reference cannot decode k = 256 blocks)
            err, back = ref.decode(enc, raw_hint=per + 64)
            ok = err == 0 and back.size == per
        t2 = time.perf_counter()
        q.put((idx, t0, t1, t2, ok))
    except Exception as e:
        q.put((idx, 0.0, 0.0, 0.0, repr(e)))


def cpu_all_cores_child(workload: str, blocksize: int) -> None:
    """`bench.py --cpu-all-cores-child W B` (started by cpu_baseline): the unmodified reference on EVERY host core at
    once, one process per core, 8 MiB of the workload each (blocks are independent: what a host-side caller
    with that many cores could do with libhuffman today).  Prints one JSON object: aggregate GiB/s = all bytes /
    (last finish - first start)."""
    import multiprocessing as mp
    try:
        usable = len(os.sched_getaffinity(0))                  # (a container may be given fewer cores than the machine has)
    except AttributeError:
        usable = os.cpu_count() or 1
    # ... or less CPU TIME than cores: a cgroup quota (cpu.max "quota period").  The GPU boxes of this pool show 256
    # hardware threads and grant 16 cores' worth of time; 256 processes then measure the throttle, not the machine
    # (8 processes 0.14, 32: 0.26, 64: 0.24, 128: 0.21, 256: 0.16 GiB/s).  Twice the quota's cores is what ran fastest.
    quota_cores = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            quota_cores = max(1, int(int(q) / int(period) + 0.5))
    except (OSError, ValueError):
        pass
    cores = min(usable, quota_cores) if quota_cores else usable
    ncpu = int(os.environ.get("BENCH_CPU_PROCESSES", "0")) or (min(usable, 2 * cores) if quota_cores and quota_cores < usable else cores)
    per = max(blocksize, (8 << 20) // blocksize * blocksize)
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(ncpu), ctx.Queue()
    procs = [ctx.Process(target=_all_cores_worker, args=(workload, blocksize, per, i, barrier, q)) for i in range(ncpu)]
    for p_ in procs:
        p_.start()
    got = [q.get(timeout=170) for _ in procs]
    for p_ in procs:
        p_.join(timeout=10)
    bad = [g for g in got if g[4] is not True]
    if bad:
        print(json.dumps({"error": "worker failed: %r" % (bad[0][4],)}))
        return
    start, mid, end = min(g[1] for g in got), max(g[2] for g in got), max(g[3] for g in got)
    total = per * ncpu
    print(json.dumps({"value": round(total / GIB / (end - start), 5), "unit": "GiB/s", "cores": cores, "processes": ncpu,
                      "hardware_threads": os.cpu_count(), "cgroup_quota_cores": quota_cores,
                      "encode_GiBps": round(total / GIB / (mid - start), 5),
                      "sample": f"{per >> 20} MiB of {workload} per process ({total >> 20} MiB), {blocksize >> 10} KiB blocks, "
                                f"encode + decode {end - start:.2f}s, all processes released together"}))


STAMP_SOURCE = ("profiles/traffic.json: TCC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE doubled) "
                "of this command, stamped for exactly these kernel sources by tools/gpu_round_profile.sh - not counted in this run")


def kernel_source_digest() -> str:
    """sha256 over the kernel sources: the stamp profiles/traffic.json carries, so that counter
    values of an older build are never printed as this build's."""
    csrc = os.path.join(ROOT, "libhuffman_amd", "csrc")
    h = hashlib.sha256()
    names = sorted(os.listdir(os.path.join(csrc, "kernels")))
    for path in [os.path.join(csrc, "kernels", f) for f in names] + [os.path.join(csrc, "hufgpu_api.hip")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(workload: str, kernel: str, n: int, bs: int):
    """HBM bytes per launch of `kernel` from the TCC counters (tools/gpu_round_profile.sh: separate
    rocprofv3 --pmc passes of this same command, FETCH_SIZE doubled per the gfx950 correction).
    Null unless the committed table was collected for exactly these kernel sources."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t.get("kernel_source_digest") != kernel_source_digest():
            return None
        if n != (1 << 30) or bs != t.get("blocksize", 65536):
            return None
        return round(t["workloads"][workload][kernel]["hbm"])
    except Exception:
        return None


def live_traffic(workload: str, kernels, passthrough=(), timeout: float = 120.0, valu: bool = False):
    """HBM bytes per launch of every kernel named in `kernels` counted in THIS run: two child passes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domain mixed in; FETCH_SIZE
    doubled: the gfx950 correction of MI355X_MICROARCH.md; KiB -> bytes); with valu=True a third pass counts SQ_INSTS_VALU
    (wave instructions per launch, under the key "__valu__").  None when rocprofv3 is not there, this
    process is itself being profiled, or a pass fails or takes too long - the stamp of profiles/traffic.json
    (same kernel sources) stays in the line then."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or any(k in os.environ for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES")) \
            or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    got = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            for ctr in ("FETCH_SIZE", "WRITE_SIZE") + (("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE") if valu else ()):
                cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(d, ctr), "-o", "p", "--",
                       sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--workload", workload,
                       "--secondary", "none", "--no-cpu-baseline", "--no-other-decode", "--no-index-free",
                       "--no-python-layer", "--no-verify", "--no-live-traffic"] + list(passthrough)
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=timeout)
                if r.returncode != 0:
                    return None
                files = glob.glob(os.path.join(d, ctr, "**", "*counter_collection.csv"), recursive=True)
                if not files:
                    return None
                total, launches = {k: 0.0 for k in kernels}, {k: set() for k in kernels}
                spans = {k: {} for k in kernels}                       # dispatch -> its nanoseconds under this pass
                with open(files[0]) as f:
                    for row in csv.DictReader(f):
                        if row["Counter_Name"] != ctr:
                            continue
                        for k in kernels:
                            if k in row["Kernel_Name"]:
                                total[k] += float(row["Counter_Value"])
                                launches[k].add(row["Dispatch_Id"])
                                try:
                                    spans[k][row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                                except (KeyError, ValueError):
                                    pass
                got[ctr] = {k: total[k] / len(launches[k]) for k in kernels if launches[k]}
                if ctr == "GRBM_GUI_ACTIVE":
                    # the clock a kernel ran at: busy cycles (rocprofv3 reports the sum over the 8 XCDs) / 8 / the dispatch's time IN THIS
                    # pass (MI355X_MICROARCH.md, DVFS give-back; reads high for dispatches of less than about 0.3 ms)
                    got["__clock__"] = {k: round(total[k] / 8.0 / sum(spans[k].values()), 3)
                                        for k in kernels if launches[k] and len(spans[k]) == len(launches[k]) and sum(spans[k].values()) > 0}
        out = {k: round(got["FETCH_SIZE"][k] * 1024 * 2 + got["WRITE_SIZE"][k] * 1024)
               for k in kernels if k in got["FETCH_SIZE"] and k in got["WRITE_SIZE"]}
        if valu and out:
            out["__valu__"] = {k: round(v) for k, v in got.get("SQ_INSTS_VALU", {}).items()}      # wave instructions per launch
            out["__clock__"] = got.get("__clock__", {})
        return out or None
    except Exception:
        return None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD process (this
    process has not touched a GPU and never will) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)                 # banners of the launcher / RCCL: not part of the result
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode


def ranks_agree(dist, torch, end_group, all_good_here: bool) -> bool:
    """The last thing the ranks of a job with more than one rank do together: does EVERY rank stand here, and did every rank's
    extra figure (root_placement) go through?  One all-reduce of a flag over a gloo group made for it (CPU tensors, a
    timeout that raises instead of a watchdog that kills): a rank that is missing or that failed makes this False on all the
    others within the group's timeout - where a barrier of the RCCL group would wait until the launcher kills the job and
    rank 0's line with it (tests/test_bench_host.py runs it with two and three gloo ranks, one failing, one missing)."""
    try:
        flag = torch.tensor([1 if all_good_here else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=end_group)
        return bool(int(flag.item()) == 1)
    except Exception:
        return False


def wait_for_file(path: str, seconds: float) -> bool:
    """(a rank that leaves with an error first gives rank 0 the time to print: the launcher tears every rank down as soon as one
    has failed)"""
    t_end = time.time() + seconds
    while time.time() < t_end:
        if os.path.exists(path):
            return True
        time.sleep(0.2)
    return False


class Bench:
    def __init__(self, args, torch, dist, codec, rank, world, local_rank, use_dist, ctl_group=None):
        self.a, self.torch, self.dist, self.codec = args, torch, dist, codec
        self.rank, self.world, self.local_rank, self.use_dist = rank, world, local_rank, use_dist
        self.dev = torch.device("cuda", local_rank)
        # the extra figure's CONTROL traffic (flags, barriers, the legs' times) goes over the gloo group of main() when there is
        # one: its timeout raises, where a barrier of the RCCL group behind a rank that failed alone waits for the watchdog
        self.ctl = ctl_group

    def ctl_barrier(self):
        if self.ctl is not None:
            self.dist.barrier(group=self.ctl)
        else:
            self.dist.barrier()

    def ctl_reduce(self, value, op):
        t = self.torch.tensor([value], dtype=self.torch.float64, device=("cpu" if self.ctl is not None else self.dev))
        self.dist.all_reduce(t, op=op, group=self.ctl)
        return float(t.item())

    def make_input(self, workload: str, n: int, first: int):
        torch = self.torch
        if workload == "logtext":
            # the text generator runs on the host: one 16 MiB tile, repeated on the device
            from libhuffman_amd import datagen
            tile = torch.from_numpy(datagen.logtext(16 << 20)).to(self.dev)
            return tile.repeat((n + tile.numel() - 1) // tile.numel())[:n].contiguous()
        data = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.codec.fill(data, workload, first=first)
        return data

    def ceilings(self, n: int) -> dict:
        """GB/s of the hand-written byte movers of kernels/fill.hpp (16 bytes per lane and access; hufgpu_calib_bandwidth)
        over n bytes on this GPU: copy = read n + write n, read = n, write (fill) = n; the best workgroup shape / cache
        policy of each, HIP events on the current stream.  What a kernel that only moves bytes reaches here - the
        measured ceilings SURVEY 8d asks to be reported beside the 8 TB/s of the data sheet.  (torch's copy_ is timed
        too: it is what rounds 1-3 called the copy ceiling.)"""
        torch = self.torch
        a = torch.empty(n, dtype=torch.uint8, device=self.dev)
        b = torch.empty(n, dtype=torch.uint8, device=self.dev)
        a.fill_(7)

        def timed(fn, reps=5):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 1e3 / reps

        out = {}
        for kind, moved in (("copy", 2.0 * n), ("read", 1.0 * n), ("fill", 1.0 * n)):
            best, which = 0.0, -1
            for v in range(self.codec.CALIB_VARIANTS):
                gbs = moved / 1e9 / timed(lambda: self.codec.calib_bandwidth(kind, v, a, b, n))
                if gbs > best:
                    best, which = gbs, v
            out[kind] = (best, which)
        out["torch_copy"] = (2.0 * n / 1e9 / timed(lambda: b.copy_(a)), -1)
        del a, b
        torch.cuda.empty_cache()
        return out

    def run(self, workload: str, steps: int, warmup: int, decode: str = None, bytes_per_gpu: int = None,
            other_decode: bool = True) -> dict:
        """K timed steps of one workload on every rank; returns rank 0's result record."""
        from libhuffman_amd.sharding import shard_range
        a, torch, dist, codec = self.a, self.torch, self.dist, self.codec
        world, rank, dev = self.world, self.rank, self.dev
        bs = a.blocksize or ((1 << 20) if workload == "logtext" else 65536)
        n_total = (bytes_per_gpu or a.bytes_per_gpu) * world
        lo, hi = shard_range(n_total, bs, rank, world)          # contiguous block range of this rank
        n = hi - lo
        nb = codec.block_count(n, bs)
        relaxed = workload == "uniform256"
        use_sub = (decode or a.decode) == "sub"

        data = self.make_input(workload, n, lo)
        out = torch.empty(codec.encode_bound(n, bs), dtype=torch.uint8, device=dev)
        offs = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        sub = codec.new_sub_index(n, bs) if use_sub else None
        RING = 8
        sizes = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(RING)]
        snaps = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(RING)]
        pending = [None] * RING
        self._k = 0

        def step():
            codec.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
            if self.use_dist:
                # the one real exchange: every rank learns where its stream starts in the job's stream.
                # Nothing on this rank's decode depends on it, so it runs on RCCL's stream beside the
                # decode.  The next encode rewrites `offs`, so the size is snapshot on the compute
                # stream first; a ring of buffers keeps the collectives of neighbouring steps apart.
                j = self._k % RING
                if pending[j] is not None:
                    pending[j].wait()
                snaps[j].copy_(offs[nb:nb + 1])
                pending[j] = dist.all_gather_into_tensor(sizes[j], snaps[j], async_op=True)
                self._k += 1
            codec.decode(out, out.numel(), offs, nb, back, relaxed=relaxed, sync=False,
                         sub_index=sub, raw_size=n, blocksize=bs)

        def drain():
            for j in range(RING):
                if pending[j] is not None:
                    pending[j].wait()
                    pending[j] = None

        for _ in range(warmup):
            step()
        drain()
        if warmup:
            codec.decode_result()
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

        # HIP events around every kernel cost ~5 us each, so inside the timed region every
        # PROFILE_EVERY-th step carries them; the per-kernel averages are over those steps
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for k in range(steps):
            codec.set_profiling(k % PROFILE_EVERY == 0, resume=k > 0)
            step()
        drain()
        ev1.record()
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        raw = codec.decode_result()
        assert raw == n, f"decode produced {raw} of {n} bytes"
        enc_prof, enc_calls = codec.profile("encode")
        dec_prof, dec_calls = codec.profile("decode")
        codec.set_profiling(False)

        per_rank_ms, allgather_ms = None, None
        if self.use_dist:
            # every rank's own clock over the K steps (the value uses the slowest), and what the step's one collective -
            # the all-gather of the compressed sizes - costs on its own (K of them back to back, after the timed region)
            mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank_ms = [round(float(t.item()) / steps * 1e3, 4) for t in every]
            elapsed = max(float(t.item()) for t in every)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                dist.all_gather_into_tensor(sizes[0], snaps[0])
            torch.cuda.synchronize()
            allgather_ms = round((time.perf_counter() - t1) / steps * 1e3, 4)

        comp_len = int(offs[nb].item())
        bit_exact = None
        if not a.no_verify:
            bit_exact = bool(torch.equal(back, data))      # full-size round trip on every rank
            if self.use_dist:
                ok = torch.tensor([1 if bit_exact else 0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                bit_exact = bool(ok.item())

        # the other decoder on the same stream, outside the timed region: what a stream without the
        # encoder's sub-index costs (and the other way round)
        other_ms = None
        if rank == 0 and use_sub and other_decode and not a.no_other_decode:
            codec.set_profiling(True)
            for _ in range(3):
                codec.decode(out, out.numel(), offs, nb, back, relaxed=relaxed, sync=False)
            codec.decode_result()
            p, c = codec.profile("decode")
            codec.set_profiling(False)
            other_ms = p["decode"] / max(c, 1)

        # ... and with no index at all (what huf_decode() gets: block discovery + probes), wall clock
        # of the call, outside the timed region; reported with the index-free record only
        raw_ms = None
        if rank == 0 and not use_sub and not a.no_other_decode:
            codec.decode_stream(out, comp_len, comp_len, back, relaxed=relaxed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                raw_res = codec.decode_stream(out, comp_len, comp_len, back, relaxed=relaxed)
            torch.cuda.synchronize()
            raw_ms = (time.perf_counter() - t0) / 3 * 1e3
            assert tuple(raw_res) == (0, n, comp_len), f"raw-stream decode returned {raw_res}"
            if not a.no_verify:
                assert torch.equal(back, data), "raw-stream decode differs from the input"

        rec = None
        if rank == 0:
            K = steps
            value = n_total * K / GIB / elapsed
            # per-launch algorithmic bytes: encode reads N writes C, decode reads C writes N
            # The sub-index is in-process side information, written by pack and read by decode_sub.  SURVEY 8d
            # counts only N and C as algorithmic bytes (side tables are implementation traffic), so `achieved`
            # does NOT include it; the line states its size so that nobody has to guess what else moves.
            sub_bytes = codec.sub_index_bytes(n, bs) if (use_sub and workload != "const41") else 0
            alg = {"pack": n + comp_len, "decode": comp_len + n, "hist256": n,
                   "tree": nb * (1024 + 2048 + 2064), "scan_sizes": nb * 24, "prepare_scan": nb * (16 + 10 + 28)}
            if workload == "const41":
                alg["pack"] = comp_len          # one-symbol blocks: the input is not read again, the payload is zeros
            # blocks below 32 KiB take the fused histogram+tree kernel: its time is reported once; from 32 KiB
            # (and below 4 MiB) on the counts and the trees are two launches, hist_lanes_kernel and tree_wave_kernel
            fused = bs < 32768
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
            if other_ms is not None:
                kernels["decode_selfsync"] = {
                    "avg_ms": round(other_ms, 4), "alg_GBps": round(alg["decode"] / 1e9 / (other_ms / 1e3), 1),
                    "note": "outside the timed region"}
            dom = max((k for k in ("pack", "decode", "hist256", "tree", "hist_tree") if k in kernels),
                      key=lambda k: kernels[k]["avg_ms"])
            achieved = alg[dom] / 1e9 / (kernels[dom]["avg_ms"] / 1e3)
            kernel_names = {"decode": "decode_sub_kernel" if use_sub else "decode_fast_kernel", "pack": "pack_kernel",
                            "hist_tree": "hist_tree_kernel", "hist256": "hist_lanes_kernel" if bs < (1 << 21) else "chunk_hist_kernel",
                            "tree": "tree_wave_kernel" if bs < (1 << 22) else "tree_kernel"}
            pipeline_bytes = 2 * (n + comp_len)     # the metric's bytes (side tables are implementation traffic)
            gpu_ms = ev0.elapsed_time(ev1) / K
            # encode-only / decode-only: this rank's bytes over the kernels of each half
            enc_ms = sum(v["avg_ms"] for k, v in kernels.items() if k in enc_prof)
            dec_ms = sum(v["avg_ms"] for k, v in kernels.items() if k in dec_prof)
            rec = {
                "value": round(value, 3),
                "ms_per_step": round(elapsed / K * 1e3, 4),
                "config": {"workload": WORKLOADS[workload], "generator": workload,
                           "bytes_per_gpu": n, "blocksize": bs, "blocks_per_gpu": nb,
                           "compressed_bytes_per_gpu": comp_len, "ratio": round(comp_len / n, 5),
                           "parallelism": f"block-sharded x{world}", "bit_exact_roundtrip": bit_exact,
                           "decoder": "sub-index (verified), one table pass per symbol" if use_sub
                                      else "self-synchronising (block index only)"},
                "roofline": {"bound": "hbm", "kernel": kernel_names[dom], "achieved": round(achieved, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                             "traffic": measured_traffic(workload, "decode_index" if (dom == "decode" and not use_sub) else dom, n, bs),
                             "alg_bytes_per_launch": alg[dom],
                             "side_bytes_per_launch": sub_bytes if dom in ("pack", "decode") else 0,
                             "pipeline_frac": round(pipeline_bytes / 1e9 / (gpu_ms / 1e3) / HBM_PEAK_GBS, 4)},
                "kernels": kernels,
                "encode_only_GiBps_per_gpu": round(n / GIB / (enc_ms / 1e3), 1) if enc_ms > 0 else None,
                "decode_only_GiBps_per_gpu": round(n / GIB / (dec_ms / 1e3), 1) if dec_ms > 0 else None,
                "gpu_ms_per_step_rank0": round(gpu_ms, 4),
                "profiled_steps": max(enc_calls, dec_calls),
            }
            if per_rank_ms is not None:
                rec["per_rank_ms_per_step"] = per_rank_ms
                rec["size_allgather_ms"] = allgather_ms
            if raw_ms is not None:
                rec["raw_stream_decode"] = {"ms": round(raw_ms, 4), "GiBps": round(n / GIB / (raw_ms / 1e3), 1),
                                            "note": "no index at all (huf_decode's input): discovery + probes, wall clock of "
                                                    "the call, outside the timed region"}
        del data, out, offs, back, sub
        torch.cuda.empty_cache()
        return rec

    def root_placement(self, workload: str, steps: int) -> dict:
        """The path with the data starting and ending on rank 0 (north_star: "RCCL scatter/gather of block buffers
        over xGMI"), through the C library's own entry points - hufgpu_encode_sharded / hufgpu_decode_sharded
        (include/huffman_gpu.h): grouped ncclSend / ncclRecv to computed offsets on a communicator of the library's
        own.  Those calls have no timeout, so they run on a thread this one gives up on after BENCH_LEG_TIMEOUT x 1.5
        seconds: the line is then printed without the figure.  If RCCL cannot be had from C (no library), the same
        movements through torch.distributed (`root_placement_torch`)."""
        import threading
        from libhuffman_amd import sharding
        a, torch, dist, codec = self.a, self.torch, self.dist, self.codec
        world, rank, dev = self.world, self.rank, self.dev
        import ctypes
        made = 1 if codec.lib.hufgpu_shard_unique_id(ctypes.create_string_buffer(128)) == 0 else 0     # can the C library reach RCCL?
        if int(self.ctl_reduce(made, dist.ReduceOp.MIN)) == 0:
            rec = self.root_placement_torch(workload, steps)
            if rec is not None:
                rec["mover"] = "torch.distributed all_to_all_single (the C library could not load RCCL)"
            return rec
        bs = a.blocksize or ((1 << 20) if workload == "logtext" else 65536)
        n_total = a.bytes_per_gpu * world
        relaxed = workload == "uniform256"
        alloc_ok, alloc_err = 1, None
        full = stream = result = None
        try:
            if rank == 0:
                full = self.make_input(workload, n_total, 0)
                stream = torch.empty(codec.encode_bound(n_total, bs), dtype=torch.uint8, device=dev)
                result = torch.empty(n_total, dtype=torch.uint8, device=dev)
        except Exception as e:
            alloc_ok, alloc_err = 0, repr(e)
        if int(self.ctl_reduce(alloc_ok, dist.ReduceOp.MIN)) == 0:
            return {"error": "allocation failed on some rank: %s" % alloc_err} if rank == 0 else None
        names = ("scatter_in", "encode", "sizes_allgather", "gather_stream", "plan", "scatter_stream", "decode", "gather_out")
        box = {"legs": {k: 0.0 for k in names}, "total": 0.0, "error": None, "stream_bytes": 0, "lens": None}

        def run():
            try:
                torch.cuda.set_device(dev)
                group = sharding.ShardGroup(codec, group=self.ctl)       # (ncclCommInitRank: every rank, or nobody comes back)
                for k in range(steps + 1):
                    if k == 1:                          # step 0 is the warm-up (RCCL sets its channels up)
                        box["legs"] = {key: 0.0 for key in names}
                        box["total"] = 0.0
                    torch.cuda.synchronize()
                    self.ctl_barrier()
                    t0 = time.perf_counter()
                    total, lens, l1 = group.encode(full, n_total, bs, stream, root=0, legs=True)
                    _, l2 = group.decode(stream, total, n_total, bs, result, root=0, own_layout=True, relaxed=relaxed, legs=True)
                    torch.cuda.synchronize()
                    self.ctl_barrier()
                    box["total"] += time.perf_counter() - t0
                    for name, ms in zip(names, list(l1) + list(l2)):
                        box["legs"][name] += ms * 1e-3
                    box["stream_bytes"], box["lens"] = total, lens
                group.close()
            except Exception as e:
                box["error"] = repr(e)

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(float(os.environ.get("BENCH_LEG_TIMEOUT", "120")) * 1.5)     # (six steps of 8 GiB each way take seconds)
        if th.is_alive():
            raise TimeoutError("hufgpu_encode_sharded / hufgpu_decode_sharded did not come back")
        if box["error"]:
            raise RuntimeError(box["error"])
        legs_all = [None] * world
        dist.all_gather_object(legs_all, {k: round(v / max(steps, 1) * 1e3, 3) for k, v in box["legs"].items()}, group=self.ctl)
        tmax_s = self.ctl_reduce(box["total"], dist.ReduceOp.MAX)
        ok = True
        if rank == 0 and not a.no_verify:
            ok = bool(torch.equal(result, full))
        if rank != 0:
            return None
        sec = tmax_s / steps
        return {"value": round(n_total / GIB / sec, 3), "unit": "GiB/s", "ms_per_step": round(sec * 1e3, 3),
                "steps": steps, "bit_exact_roundtrip": ok, "stream_bytes": int(box["stream_bytes"]),
                "mover": "hufgpu_encode_sharded + hufgpu_decode_sharded (C ABI: grouped ncclSend/ncclRecv on the library's own communicator)",
                "legs_ms_rank0": {key: round(v / steps * 1e3, 3) for key, v in box["legs"].items()},
                "legs_ms_per_rank": legs_all,
                "note": "input and output live on rank 0; every leg is synchronised (no overlap between legs); decode with "
                        "the block index and sub-index every rank kept from the encode (HUFGPU_SHARD_OWN_LAYOUT)"}

    def root_placement_torch(self, workload: str, steps: int) -> dict:
        """The same movements through torch.distributed: scatter input shards -> encode -> gather compressed shards
        (+ the size all-gather that places them) -> scatter them again -> decode -> gather the output.  Each
        movement is ONE variable-size all-to-all (grouped send/recv inside RCCL; RCCL has no gatherv)."""
        from libhuffman_amd import sharding
        a, torch, dist, codec = self.a, self.torch, self.dist, self.codec
        world, rank, dev = self.world, self.rank, self.dev
        bs = a.blocksize or ((1 << 20) if workload == "logtext" else 65536)
        n_total = a.bytes_per_gpu * world
        plan = sharding.shard_plan(n_total, bs, world)
        lo, hi = plan[rank]
        n = hi - lo
        nb = codec.block_count(n, bs)
        relaxed = workload == "uniform256"
        # every rank allocates first, then all agree that everybody could: a rank that fails alone
        # (rank 0 holds the whole job's input, stream and output) must not leave the others in a collective
        alloc_ok, alloc_err = 1, None
        try:
            full = self.make_input(workload, n_total, 0) if rank == 0 else None
            shard = torch.empty(n, dtype=torch.uint8, device=dev)
            out = torch.empty(codec.encode_bound(n, bs), dtype=torch.uint8, device=dev)
            offs = torch.empty(nb + 1, dtype=torch.int64, device=dev)
            sub = codec.new_sub_index(n, bs)
            back = torch.empty(n, dtype=torch.uint8, device=dev)
            result = torch.empty(n_total if rank == 0 else 0, dtype=torch.uint8, device=dev)
            spare = torch.empty(codec.encode_bound(n_total, bs) if rank == 0 else 0, dtype=torch.uint8, device=dev)
            del spare                                   # what the gathered stream will need on rank 0
        except Exception as e:
            alloc_ok, alloc_err = 0, repr(e)
        if int(self.ctl_reduce(alloc_ok, dist.ReduceOp.MIN)) == 0:
            return {"error": "allocation failed on some rank: %s" % alloc_err} if rank == 0 else None
        in_sizes = [h - l for l, h in plan]
        gathered = None
        leg_timeout = float(os.environ.get("BENCH_LEG_TIMEOUT", "120"))     # seconds a movement may take
        legs = {"scatter_in": 0.0, "encode": 0.0, "gather_stream": 0.0, "scatter_stream": 0.0, "decode": 0.0,
                "gather_out": 0.0}

        def timed(name, fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            legs[name] += time.perf_counter() - t
            return r

        total = 0.0
        for k in range(steps + 1):
            if k == 1:                                  # step 0 is the warm-up (RCCL sets its channels up)
                legs = {key: 0.0 for key in legs}
                total = 0.0
            torch.cuda.synchronize()
            self.ctl_barrier()
            t0 = time.perf_counter()
            timed("scatter_in", lambda: sharding.scatter_from_root(full, in_sizes, shard, 0, timeout=leg_timeout))
            timed("encode", lambda: codec.encode(shard, bs, out=out, offsets=offs, sync=False, sub_index=sub))
            clen = int(offs[nb].item())
            gathered, csizes = timed("gather_stream", lambda: sharding.gatherv_to_root(out, clen, 0, timeout=leg_timeout))
            timed("scatter_stream", lambda: sharding.scatter_from_root(gathered, csizes, out[:clen], 0, timeout=leg_timeout))
            timed("decode", lambda: codec.decode(out, clen, offs, nb, back, relaxed=relaxed, sync=True,
                                                 sub_index=sub, raw_size=n, blocksize=bs))
            timed("gather_out", lambda: sharding.gather_to_root(back, in_sizes, result, 0, timeout=leg_timeout))
            torch.cuda.synchronize()
            self.ctl_barrier()
            total += time.perf_counter() - t0
        legs_all = [None] * world
        dist.all_gather_object(legs_all, {k: round(v / max(steps, 1) * 1e3, 3) for k, v in legs.items()}, group=self.ctl)
        tmax_s = self.ctl_reduce(total, dist.ReduceOp.MAX)
        ok = True
        if rank == 0 and not a.no_verify:
            ok = bool(torch.equal(result, full))
        if rank != 0:
            return None
        sec = tmax_s / steps
        return {"value": round(n_total / GIB / sec, 3), "unit": "GiB/s", "ms_per_step": round(sec * 1e3, 3),
                "steps": steps, "bit_exact_roundtrip": ok, "stream_bytes": int(sum(csizes)),
                "legs_ms_rank0": {key: round(v / steps * 1e3, 3) for key, v in legs.items()},
                "legs_ms_per_rank": legs_all,
                "note": "input and output live on rank 0; every leg is synchronised (no overlap between legs)"}


C_API_SIZES = (10, 4 << 10, 64 << 10, 1 << 20, 64 << 20, 1 << 30)


def c_api_by_size(lib, sizes, blocksize: int, budget_s: float = 6.0, relaxed_setter=None) -> dict:
    """huf_encode() + huf_decode() on huf_memopen streams (the drop-in boundary itself: host bytes in, host bytes out,
    PCIe included) for inputs of `sizes` bytes of zipf255 - microseconds per call and GiB/s of the pair.  `lib` is any
    library with libhuffman's ABI (include/huffman.h): this build's, or - in the cpu_baseline leg - the unmodified
    reference's (oracle/_ref), which is why the streams are driven through ctypes here and not through a wrapper."""
    import ctypes as C
    import numpy as np
    from libhuffman_amd import _native as N, datagen
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    lib.huf_memopen.argtypes = [C.POINTER(C.POINTER(N.ReadWriter)), C.POINTER(C.c_void_p), C.c_size_t]
    lib.huf_memclose.argtypes = [C.POINTER(C.POINTER(N.ReadWriter))]
    lib.huf_memlen.argtypes = [C.POINTER(N.ReadWriter), C.POINTER(C.c_size_t)]
    lib.huf_encode.argtypes = [C.POINTER(N.Config)]
    lib.huf_decode.argtypes = [C.POINTER(N.Config)]
    tile = datagen.zipf255(min(max(sizes), 16 << 20))
    out = {}
    for n in sizes:
        data = tile[:n] if n <= tile.size else np.tile(tile, (n + tile.size - 1) // tile.size)[:n]
        data = np.ascontiguousarray(data)
        t_enc = t_dec = 0.0
        each = []                                      # (encode, decode) seconds of every timed call
        calls, ok = 0, True
        while calls < 3 or (t_enc + t_dec < budget_s / len(sizes) and calls < 2000):
            rin, rout, rback = C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)()
            bin_, bout, bback = C.c_void_p(), C.c_void_p(), C.c_void_p()
            assert lib.huf_memopen(C.byref(rin), C.byref(bin_), max(n, 1)) == 0
            assert lib.huf_memopen(C.byref(rout), C.byref(bout), max(n + n // 4, 4096)) == 0
            assert lib.huf_memopen(C.byref(rback), C.byref(bback), max(n, 1)) == 0
            assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
            cfg = N.Config(n, blocksize, 0, 0, rin, rout)
            t0 = time.perf_counter()
            e1 = lib.huf_encode(C.byref(cfg))
            t1 = time.perf_counter()
            m = C.c_size_t()
            lib.huf_memlen(rout, C.byref(m))
            dcfg = N.Config(m.value, 0, 0, 0, rout, rback)
            t2 = time.perf_counter()
            e2 = lib.huf_decode(C.byref(dcfg))
            t3 = time.perf_counter()
            lib.huf_memlen(rback, C.byref(m))
            good = e1 == 0 and e2 == 0 and m.value == n and C.string_at(bback.value, min(n, 4096)) == data[:4096].tobytes()
            for r in (rin, rout, rback):
                lib.huf_memclose(C.byref(r))
            for b in (bin_, bout, bback):
                libc.free(b)
            # the first call of a size warms buffers and pages and is not counted - unless a call takes seconds (the
            # reference on one core from 64 MiB on: two calls, both counted, the warm-up is nothing beside them)
            if calls == 0:
                skip_first = n < (64 << 20) or (t1 - t0) + (t3 - t2) < 0.5
            if calls >= 1 or not skip_first:
                t_enc += t1 - t0
                t_dec += t3 - t2
                each.append((t1 - t0, t3 - t2))
            ok = ok and good
            calls += 1
            if n >= (64 << 20) and calls >= (6 if skip_first else 2):
                break
        timed = calls - 1 if skip_first else calls
        rec = {"encode_us": round(t_enc / timed * 1e6, 1), "decode_us": round(t_dec / timed * 1e6, 1),
               "GiBps": round(n * timed / GIB / (t_enc + t_dec), 5), "calls": timed, "roundtrip_ok": bool(ok)}
        if n >= (64 << 20) and skip_first:
            # Calls of tens of milliseconds on a shared host: one call in five meets a stall of the system's (a buffer that
            # grows, pages compacted) that is several times a call.  The figure is the MEDIAN of the five timed calls; the
            # mean and the slowest call stand beside it.
            enc_sorted, dec_sorted = sorted(e for e, _ in each), sorted(d for _, d in each)
            med_e, med_d = enc_sorted[len(each) // 2], dec_sorted[len(each) // 2]
            rec.update({"encode_us": round(med_e * 1e6, 1), "decode_us": round(med_d * 1e6, 1),
                        "GiBps": round(n / GIB / (med_e + med_d), 5), "statistic": "median of %d calls" % len(each),
                        "mean_us": [round(t_enc / timed * 1e6, 1), round(t_dec / timed * 1e6, 1)],
                        "slowest_us": [round(enc_sorted[-1] * 1e6, 1), round(dec_sorted[-1] * 1e6, 1)]})
        out[str(n)] = rec
    return out


def huffmanfile_layer(n: int, blocksize: int, reps: int = 2) -> dict:
    """BASELINE.json configs[4]'s shape on this GPU: synthetic log text through the Python layer
    (libhuffman_amd.huffmanfile.compress / decompress = huf_encode / huf_decode behind memstreams), host
    bytes in, host bytes out.  PCIe-inclusive by construction: reported beside the resident-data figure,
    never instead of it."""
    import numpy as np
    from libhuffman_amd import datagen, huffmanfile
    tile = datagen.logtext(16 << 20)
    data = np.tile(tile, (n + tile.size - 1) // tile.size)[:n].tobytes()
    comp = huffmanfile.compress(data, blocksize)              # warm-up: sessions, pinned buffers, page faults
    back = huffmanfile.decompress(comp)
    ok = back == data
    del back
    # What is timed are the CALLS.  The result of the call before is dropped OUTSIDE the timed windows: on this box giving a
    # gigabyte back to the system takes as long as a call (40-60 ms: tools/time_host_link.py), and a loop that
    # rebinds `comp = compress(...)` pays that inside the statement it times - round 3's and round 4's earlier lines did (6.0
    # GiB/s where the calls alone give 10-11).  The cost is reported beside the rates.
    t_c = t_d = t_free = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        fresh = huffmanfile.compress(data, blocksize)
        t1 = time.perf_counter()
        del comp
        comp = fresh
        del fresh
        t2 = time.perf_counter()
        back = huffmanfile.decompress(comp)
        t3 = time.perf_counter()
        ok = ok and back == data
        t4 = time.perf_counter()
        del back
        t_free += (t2 - t1) + (time.perf_counter() - t4)
        t_c += t1 - t0
        t_d += t3 - t2
    return {"value": round(n * reps / GIB / (t_c + t_d), 3), "unit": "GiB/s",
            "workload": "configs[4] per-GPU share: %d MiB of synthetic log text, blocksize=%d KiB, through "
                        "huffmanfile.compress/decompress (host bytes in and out)" % (n >> 20, blocksize >> 10),
            "compress_GiBps": round(n * reps / GIB / t_c, 3), "decompress_GiBps": round(n * reps / GIB / t_d, 3),
            "dropping_the_results_ms_per_pair": round(t_free / reps * 1e3, 1),
            "value_with_the_results_dropped_inside": round(n * reps / GIB / (t_c + t_d + t_free), 3),
            "ratio": round(len(comp) / n, 5), "bit_exact_roundtrip": bool(ok), "reps": reps}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="zipf255", choices=sorted(WORKLOADS))
    ap.add_argument("--secondary", default=None,
                    help="comma-separated workloads reported under 'secondary' (default: uniform256,const41 "
                         "beside zipf255; 'none' = only the headline workload)")
    ap.add_argument("--decode", default="sub", choices=["sub", "selfsync"],
                    help="sub = with the encoder's sub-index (verified on the device); selfsync = block index only")
    ap.add_argument("--bytes-per-gpu", type=int, default=1 << 30)
    ap.add_argument("--blocksize", type=int, default=None, help="default 64 KiB (1 MiB for logtext)")
    ap.add_argument("--placement", default="auto", choices=["auto", "resident", "root"],
                    help="root = also time the scatter/gather form (default when more than one rank runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-other-decode", action="store_true")
    ap.add_argument("--no-index-free", action="store_true", help="skip the timed loop with the block index alone")
    ap.add_argument("--no-python-layer", action="store_true", help="skip the huffmanfile (configs[4] shape) figure")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not count the longest kernel's HBM bytes with two rocprofv3 --pmc child passes (the stamp of profiles/traffic.json stays)")
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-all-cores-child":
        cpu_all_cores_child(sys.argv[2], int(sys.argv[3]))
        return
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

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
    end_group = None
    done_path = os.path.join(tempfile.gettempdir(), "huf_bench_done_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "0")))
    if world > 1:
        import datetime
        if rank == 0 and os.path.lexists(done_path):
            try:                                     # (a job of the same port that did not end clean; only a plain file of ours)
                st = os.lstat(done_path)
                import stat
                if stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid():
                    os.remove(done_path)
            except OSError:
                pass
        end_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("BENCH_END_TIMEOUT", "180"))))

    from libhuffman_amd.codec import GpuCodec

    codec = GpuCodec(local_rank)
    bench = Bench(args, torch, dist, codec, rank, world, local_rank, use_dist, ctl_group=end_group)
    if args.secondary is None:
        secondary = DEFAULT_SECONDARY.get(args.workload, [])
    else:
        secondary = [w for w in args.secondary.split(",") if w and w != "none"]
    for w in secondary:
        if w not in WORKLOADS:
            raise SystemExit(f"unknown secondary workload {w}")

    main_rec = bench.run(args.workload, args.steps, args.warmup)
    sec_recs = {w: bench.run(w, args.steps, args.warmup) for w in secondary}
    # the same K steps with the BLOCK INDEX ALONE (no sub-index: what a stream that comes without the
    # encoder's side information costs) - a timed region of its own, so that the driver's record holds both
    if args.decode == "sub" and not args.no_index_free:
        sec_recs[args.workload + "_index_free"] = bench.run(args.workload, args.steps, args.warmup, decode="selfsync",
                                                           other_decode=False)
    # BASELINE.json configs[3] is 16 GiB of uniform bytes over 8 GPUs: at world 8 that is 2 GiB per rank
    if world == 8 and "uniform256" in secondary and args.bytes_per_gpu == (1 << 30):
        sec_recs["uniform256_16GiB"] = bench.run("uniform256", max(2, args.steps // 2), min(args.warmup, 2),
                                                 bytes_per_gpu=2 << 30, other_decode=False)
        if rank == 0:
            sec_recs["uniform256_16GiB"]["config"]["workload"] = \
                "configs[3]: 16 GiB uniform-random bytes (uniform256 seed 1), blocksize=64KiB, sharded across 8 MI355X"
    ceil = bench.ceilings(1 << 30) if rank == 0 else None
    copy_gbs = ceil["copy"][0] if ceil else None
    # The scatter/gather form is an extra figure.  Every movement in it is waited for with a deadline
    # (BENCH_LEG_TIMEOUT), so a rank that fails alone shows as an exception here, not as a hang: the line is
    # then printed without the figure and the ranks leave without the final barrier.
    root_rec, root_ok = None, True
    if use_dist and args.placement in ("auto", "root"):
        try:
            root_rec = bench.root_placement(args.workload, max(2, min(args.steps, 5)))
        except Exception as e:                      # the extra figure never takes the headline down with it
            root_rec, root_ok = {"error": repr(e)}, False
    def host_legs():
        """the two figures through the host API (rank 0 only): the Python layer on configs[4]'s shape and the C API by
        input size.  With more than one rank they run AFTER the other ranks have left, over every GPU of the node
        (HUF_GPU_DEVICES=all: one huffmanfile call dealt out over the sessions - configs[4]'s stated route)."""
        py, capi = None, None
        if args.no_python_layer:
            return py, capi
        try:
            n_py = 1 << 30
            if world > 1:
                os.environ["HUF_GPU_DEVICES"] = "all"        # (read when the host API opens its first session: not before this)
                n_py = min(world, 4) << 30
            py = huffmanfile_layer(n_py, 1 << 20)
            if world > 1:
                import ctypes as C
                from libhuffman_amd import _native as N
                fe, fd, conf = C.c_int(0), C.c_int(0), C.c_int(0)
                N.load().huf_gpu_fanouts(C.byref(fe), C.byref(fd))
                live = N.load().huf_gpu_sessions(C.byref(conf))
                py["sessions"] = {"configured": conf.value, "live": live, "fanout_encodes": fe.value, "fanout_decodes": fd.value}
        except Exception as e:
            py = {"error": repr(e)}
        try:
            from libhuffman_amd import _native as N
            capi = {"what": "huf_encode + huf_decode on huf_memopen streams, zipf255, blocksize 64 KiB: host bytes in and out "
                            "(PCIe and every launch included); microseconds per call",
                    "by_bytes": c_api_by_size(N.load(), C_API_SIZES, 65536)}
        except Exception as e:
            capi = {"error": repr(e)}
        return py, capi

    py_rec, capi_rec = (None, None)
    if rank == 0 and world == 1:
        py_rec, capi_rec = host_legs()

    result = None
    if rank == 0:
        bs = main_rec["config"]["blocksize"]
        result = {
            "metric": "encode+decode GiB/s (uncompressed) on %s blocks" % ("64KiB" if bs == 65536 else "%dKiB" % (bs >> 10)),
            "value": main_rec["value"],
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_rec["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
        }
        result.update({k: v for k, v in main_rec.items() if k not in ("value", "ms_per_step")})
        result["secondary"] = {
            w: {"value": r["value"], "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                "ratio": r["config"]["ratio"], "bit_exact_roundtrip": r["config"]["bit_exact_roundtrip"],
                "roofline": {"kernel": r["roofline"]["kernel"], "frac": r["roofline"]["frac"],
                             "achieved": r["roofline"]["achieved"], "traffic": r["roofline"]["traffic"],
                             "traffic_source": (STAMP_SOURCE if r["roofline"]["traffic"] is not None else None),
                             "pipeline_frac": r["roofline"]["pipeline_frac"]},
                "kernels": {k: v["avg_ms"] for k, v in r["kernels"].items()},
                **({"raw_stream_decode": r["raw_stream_decode"]} if "raw_stream_decode" in r else {})}
            for w, r in sec_recs.items()}
        # the reference's OWN boundary (src/decoder.c:201-287 takes the stream and nothing else): encode + the decode of a
        # stream that comes with no index at all, against the same 8 TB/s
        idx_rec = sec_recs.get(args.workload + "_index_free")
        if idx_rec is not None and "raw_stream_decode" in idx_rec:
            enc_ms = sum(v["avg_ms"] for k, v in main_rec["kernels"].items() if k in ("hist256", "tree", "scan_sizes", "pack", "hist_tree"))
            raw_ms = idx_rec["raw_stream_decode"]["ms"]
            n1 = main_rec["config"]["bytes_per_gpu"]
            c1 = main_rec["config"]["compressed_bytes_per_gpu"]
            gbs = 2 * (n1 + c1) / 1e9 / ((enc_ms + raw_ms) / 1e3)
            result["roofline"]["ref_boundary"] = {
                "what": "2 (N + C) bytes over encode + raw-stream decode (huf_decode's input: no block index, no sub-index), one GPU",
                "encode_ms": round(enc_ms, 4), "raw_stream_decode_ms": round(raw_ms, 4), "achieved": round(gbs, 1),
                "frac": round(gbs / HBM_PEAK_GBS, 4), "GiBps": round(n1 / GIB / ((enc_ms + raw_ms) / 1e3), 1)}
        if root_rec is not None:
            result["root_placement"] = root_rec
        if py_rec is not None:
            result["secondary"]["logtext_huffmanfile"] = py_rec
        if capi_rec is not None:
            result["secondary"]["c_api_memstream"] = capi_rec
        result["roofline"]["traffic_source"] = STAMP_SOURCE if result["roofline"].get("traffic") is not None else None
        if world == 1 and not args.no_live_traffic:
            same = ["--bytes-per-gpu", str(args.bytes_per_gpu), "--decode", args.decode]
            if args.blocksize is not None:
                same += ["--blocksize", str(args.blocksize)]
            live_src = ("counted in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child passes "
                        "of this command (FETCH_SIZE doubled: gfx950)")
            want = sorted({result["roofline"]["kernel"], "pack_kernel", "hist_lanes_kernel", "tree_wave_kernel"})
            lt = live_traffic(args.workload, want, same, valu=True)
            valu = lt.pop("__valu__", None) if lt else None
            clocks = lt.pop("__clock__", None) if lt else None
            if lt and result["roofline"]["kernel"] in lt:
                result["roofline"]["traffic"] = lt[result["roofline"]["kernel"]]
                result["roofline"]["traffic_source"] = live_src
            if lt:
                result["kernel_traffic"] = {"bytes_per_launch": lt, "source": live_src}
            if valu:
                # how much of a kernel's time its vector instructions alone take to ISSUE: a wave instruction holds its SIMD
                # for 4 cycles, the chip has 256 CUs x 4 SIMDs at SIMD_CLOCK_HZ (the peak engine clock: a lower bound of the share)
                kms = {"pack_kernel": "pack", "hist_lanes_kernel": "hist256", "tree_wave_kernel": "tree",
                       "decode_sub_kernel": "decode", "decode_fast_kernel": "decode"}
                fr = {k: round(v * 4.0 / (1024 * SIMD_CLOCK_HZ * main_rec["kernels"][kms[k]]["avg_ms"] * 1e-3), 4)
                      for k, v in valu.items() if k in kms and kms[k] in main_rec["kernels"] and main_rec["kernels"][kms[k]]["avg_ms"] > 0}
                result["kernel_valu"] = {"wave_instructions_per_launch": valu, "issue_frac": fr,
                                         "source": "counted in this run: rocprofv3 --pmc SQ_INSTS_VALU, a child pass of this command; "
                                                   "issue_frac = instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x the kernel's time)"}
                if result["roofline"]["kernel"] in fr:
                    result["roofline"]["valu_issue_frac"] = fr[result["roofline"]["kernel"]]
                if clocks:
                    # the same share at the clock the kernel was seen to run at (GRBM_GUI_ACTIVE / 8 / the dispatch's time in a
                    # counter pass of its own; the guide: reads high below ~0.3 ms a dispatch, profiled passes clock a few % lower)
                    fr_m = {k: round(v * 4.0 / (1024 * clocks[k] * 1e9 * main_rec["kernels"][kms[k]]["avg_ms"] * 1e-3), 4)
                            for k, v in valu.items() if k in fr and clocks.get(k)}
                    result["kernel_valu"]["clock_GHz"] = clocks
                    result["kernel_valu"]["issue_frac_at_measured_clock"] = fr_m
                    result["kernel_valu"]["clock_source"] = ("rocprofv3 --pmc GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / the dispatch's own time, "
                                                             "a child pass of this command")
                    if result["roofline"]["kernel"] in clocks:
                        result["roofline"]["clock_GHz"] = clocks[result["roofline"]["kernel"]]
            # the index-alone decoder's kernel: one more pair of passes of the same command with --decode selfsync
            if (args.workload + "_index_free") in sec_recs and args.decode == "sub":
                same_idx = [x if x != args.decode else "selfsync" for x in same]
                li = live_traffic(args.workload, ["decode_fast_kernel"], same_idx)
                if li:
                    result.setdefault("kernel_traffic", {"bytes_per_launch": {}, "source": live_src})["bytes_per_launch"].update(li)
                    idx = result["secondary"][args.workload + "_index_free"]["roofline"]
                    if idx.get("kernel") == "decode_fast_kernel":          # the live count in the record it belongs to, not the stamp
                        idx["traffic"], idx["traffic_source"] = li["decode_fast_kernel"], live_src
        result["roofline"]["copy_ceiling_GBps"] = round(copy_gbs, 1)
        result["roofline"]["read_ceiling_GBps"] = round(ceil["read"][0], 1)
        result["roofline"]["write_ceiling_GBps"] = round(ceil["fill"][0], 1)
        result["roofline"]["torch_copy_GBps"] = round(ceil["torch_copy"][0], 1)
        result["roofline"]["ceiling_source"] = ("hufgpu_calib_bandwidth: hand-written 16-byte-per-lane copy / read / fill kernels "
                                                "(kernels/fill.hpp), best of %d shapes each over 1 GiB: variants %d / %d / %d"
                                                % (bench.codec.CALIB_VARIANTS, ceil["copy"][1], ceil["read"][1], ceil["fill"][1]))
        result["roofline"]["frac_of_copy_ceiling"] = round(result["roofline"]["achieved"] / copy_gbs, 4)
        if world > 1:
            result["multi_gpu_note"] = ("ranks are block-sharded; value = all ranks' bytes over the slowest rank's time; "
                                        "cpu_baseline is rank 0's host")
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload, bs)
            try:                                         # the reference on the sizes of secondary.c_api_memstream (bounded: up to 64 MiB)
                import ctypes
                from oracle.oracle import REF_SO, Reference
                if Reference.available():
                    result["cpu_baseline"]["c_api_by_bytes"] = c_api_by_size(ctypes.CDLL(REF_SO), [z for z in C_API_SIZES if z <= (64 << 20)],
                                                                                65536, budget_s=4.0)
            except Exception as e:
                result["cpu_baseline"]["c_api_by_bytes"] = {"error": repr(e)}

    # RCCL writes its version banner through C stdio, which a pipe only sees when a process exits:
    # every rank pushes its buffer out before the last barrier, so that rank 0's JSON line is the
    # last line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if world > 1:
        # The ranks part WITHOUT a barrier of the RCCL group and without its teardown (either waits for a rank that is gone until
        # the launcher kills the job, and the line with it): they agree over the gloo group whether all of them got here clean.
        all_ok = ranks_agree(dist, torch, end_group, root_ok)
        if all_ok:
            # every rank stands here: the RCCL group can be taken down the ordinary way, by all of them at once (rank 0 goes on to
            # use the GPUs alone: no communicator, and no watchdog of one, must be left behind that could mind the others' leaving)
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        if rank != 0:
            if not all_ok:
                wait_for_file(done_path, 240.0)      # (rank 0 prints at once when the ranks do not agree)
            os._exit(0 if all_ok else 1)
        if all_ok:
            py_rec, capi_rec = host_legs()           # the other ranks are idle or gone: every GPU of the node is this process's
            if py_rec is not None:
                result["secondary"]["logtext_huffmanfile"] = py_rec
            if capi_rec is not None:
                result["secondary"]["c_api_memstream"] = capi_rec
        else:
            result["ranks_left_clean"] = False       # (the error, if it was this rank's, is inside root_placement)
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)
        try:
            os.close(os.open(done_path, os.O_CREAT | os.O_EXCL | os.O_WRONLY | os.O_NOFOLLOW, 0o600))
        except OSError:
            pass
        os._exit(0 if all_ok else 1)
    if not root_ok:
        if rank == 0:
            print(json.dumps(result), flush=True)
        os._exit(1)                                  # (the line is there, with the error inside root_placement; the job did not run clean)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
