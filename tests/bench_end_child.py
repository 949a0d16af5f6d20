"""One rank of tests/test_bench_host.py::test_the_ranks_of_a_job_agree_at_its_end (gloo, CPU).
usage: bench_end_child.py rank world port case out_dir"""
import datetime, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench

rank, world, port, case, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
end_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=6))
good_here = not (case == "one_fails" and rank == 1)
if case == "one_missing" and rank == world - 1:
    time.sleep(1.0)
    os._exit(3)                                      # this rank never gets to the end
t0 = time.time()
agreed = bench.ranks_agree(dist, torch, end_group, good_here)
with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
    f.write("%d %.1f" % (1 if agreed else 0, time.time() - t0))
os._exit(0)
