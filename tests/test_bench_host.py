"""Host-side pieces of bench.py that need no GPU: where `roofline.traffic` comes from.

The figure is counted in the run itself (two child passes under `rocprofv3 --pmc`, bench.live_traffic) and falls
back to the table of profiles/traffic.json, which is only believed when it was collected for exactly the kernel
sources of the tree (bench.measured_traffic).  The GPU side of it: tests/test_gpu_parity.py
(test_bench_counts_the_longest_kernels_traffic_in_its_own_run)."""
import json
import os
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PROFILER_ENV = ("ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES")


@pytest.fixture
def clean_env(monkeypatch):
    for k in PROFILER_ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LD_PRELOAD", "")
    return monkeypatch


def fake_profiler(tmp_path, body):
    """an executable named rocprofv3 that understands `--pmc CTR ... -d DIR -o NAME -- cmd...`"""
    d = tmp_path / "bin"
    d.mkdir()
    exe = d / "rocprofv3"
    exe.write_text("#!/usr/bin/env python3\nimport os, sys\na = sys.argv[1:]\nctr = a[a.index('--pmc') + 1]\n"
                   "out = a[a.index('-d') + 1]\nname = a[a.index('-o') + 1]\ncmd = a[a.index('--') + 1:]\n" + body)
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    return str(d)


def test_the_stamped_table_is_only_believed_for_the_sources_it_was_collected_for(monkeypatch):
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    assert len(t["kernel_source_digest"]) == 16 and "zipf255" in t["workloads"]
    monkeypatch.setattr(bench, "kernel_source_digest", lambda: t["kernel_source_digest"])
    v = bench.measured_traffic("zipf255", "decode", 1 << 30, 65536)
    assert v == round(t["workloads"]["zipf255"]["decode"]["hbm"]) and v > (1 << 30)
    assert bench.measured_traffic("zipf255", "decode", 1 << 29, 65536) is None        # another size
    assert bench.measured_traffic("zipf255", "decode", 1 << 30, 1 << 20) is None      # another block size
    assert bench.measured_traffic("zipf255", "no_such_kernel", 1 << 30, 65536) is None
    monkeypatch.setattr(bench, "kernel_source_digest", lambda: "0" * 16)
    assert bench.measured_traffic("zipf255", "decode", 1 << 30, 65536) is None        # other kernel sources


def test_the_digest_covers_every_kernel_source():
    a = bench.kernel_source_digest()
    assert len(a) == 16 and a == bench.kernel_source_digest()


def test_no_counter_passes_without_a_profiler_or_under_one(clean_env, tmp_path):
    clean_env.setenv("PATH", str(tmp_path))                       # no rocprofv3 there
    assert bench.live_traffic("zipf255", ["decode_sub_kernel"]) is None
    clean_env.setenv("PATH", fake_profiler(tmp_path, "sys.exit(3)\n") + os.pathsep + os.environ["PATH"])
    assert bench.live_traffic("zipf255", ["decode_sub_kernel"]) is None           # a pass that fails
    clean_env.setenv("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so")
    assert bench.live_traffic("zipf255", ["decode_sub_kernel"]) is None           # this process is being profiled


def test_counter_passes_are_read_per_launch_of_the_named_kernel(clean_env, tmp_path):
    """FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE counts twice (gfx950); the average over the kernel's launches;
    the child command is this script with the extra work switched off and the parent's size handed down"""
    body = (
        "assert cmd[1].endswith('bench.py') and '--no-live-traffic' in cmd and '--bytes-per-gpu' in cmd, cmd\n"
        "os.makedirs(out, exist_ok=True)\n"
        "rows = ['Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value']\n"
        "val = {'FETCH_SIZE': 1000.0, 'WRITE_SIZE': 500.0}[ctr]\n"
        "for disp in (7, 9):\n"
        "    rows.append('%d,\"void hufgpu::decode_sub_kernel<512>(unsigned char const*)\",%s,%f' % (disp, ctr, val / 2))\n"
        "    rows.append('%d,\"void hufgpu::decode_sub_kernel<512>(unsigned char const*)\",%s,%f' % (disp, ctr, val / 2))\n"
        "rows.append('11,\"void hufgpu::pack_kernel<256, true>(unsigned char const*)\",%s,99999' % ctr)\n"
        "open(os.path.join(out, name + '_counter_collection.csv'), 'w').write('\\n'.join(rows) + '\\n')\n")
    clean_env.setenv("PATH", fake_profiler(tmp_path, body) + os.pathsep + os.environ["PATH"])
    got = bench.live_traffic("zipf255", ["decode_sub_kernel"], ["--bytes-per-gpu", str(1 << 28)])
    assert got == {"decode_sub_kernel": 1000 * 1024 * 2 + 500 * 1024}
    # several kernels out of the same two passes (round 4: pack_kernel and decode_fast_kernel are counted too)
    got = bench.live_traffic("zipf255", ["decode_sub_kernel", "pack_kernel", "no_such_kernel"], ["--bytes-per-gpu", str(1 << 28)])
    assert got == {"decode_sub_kernel": 1000 * 1024 * 2 + 500 * 1024, "pack_kernel": 99999 * 1024 * 3}
    assert bench.live_traffic("zipf255", ["no_such_kernel"]) is None


@pytest.mark.parametrize("case,world,want", [("all_good", 3, True), ("one_fails", 3, False), ("one_missing", 2, False)])
def test_the_ranks_of_a_job_agree_at_its_end(tmp_path, case, world, want):
    """bench.py with more than one rank ends with bench.ranks_agree over a gloo group instead of a barrier and a teardown of the RCCL
    group: every rank learns within the group's timeout whether all of them stand at the end clean - also when one of them failed
    its extra figure, also when one is gone - and rank 0 prints its line either way."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_end_child.py"), str(r), str(world), port, case, str(tmp_path)])
             for r in range(world)]
    for p in procs:
        p.wait(timeout=120)
    present = world - (1 if case == "one_missing" else 0)
    for r in range(present):
        agreed, seconds = (tmp_path / ("rank%d" % r)).read_text().split()
        assert (agreed == "1") == want, (case, r)
        assert float(seconds) < 30.0
