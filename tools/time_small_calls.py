"""Microseconds per huf_encode()/huf_decode() call on memory streams by input size, this build against the reference
(oracle/_ref, CPU) - the drop-in boundary at the sizes the reference's own tests use (test/encode_test.c:12-45)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from libhuffman_amd import _native as N
sizes = [1, 10, 1000, 4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 16 << 20]
ours = bench.c_api_by_size(N.load(), sizes, 65536, budget_s=8.0)
print("this build :", {k: (v["encode_us"], v["decode_us"], v["roundtrip_ok"]) for k, v in ours.items()})
try:
    from oracle.oracle import REF_SO
    ref = bench.c_api_by_size(ctypes.CDLL(REF_SO), sizes, 65536, budget_s=8.0)
    print("reference  :", {k: (v["encode_us"], v["decode_us"]) for k, v in ref.items()})
    print("ratio (ours / reference, encode+decode):", {k: round((ours[k]["encode_us"] + ours[k]["decode_us"]) / (ref[k]["encode_us"] + ref[k]["decode_us"]), 2) for k in ours})
except Exception as e:
    print("reference not available:", e)
