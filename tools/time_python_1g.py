"""bench.py's Python-layer leg alone (1 GiB of log text in 1 MiB blocks through huffmanfile), optionally by HUF_GPU_COPY_LANES."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lanes in (sys.argv[1:] or ["default"]):
    env = dict(os.environ)
    if lanes != "default":
        env["HUF_GPU_COPY_LANES"] = lanes
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench, json; print(json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))" % ROOT],
                       env=env, capture_output=True, text=True, timeout=900)
    print("lanes", lanes, (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)
