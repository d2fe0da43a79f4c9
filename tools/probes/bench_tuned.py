"""bench.py under hf_debug_set_tuning(bits) - A/B of dispatch options that have no environment switch (per-thread library state):
python tools/probes/bench_tuned.py <bits> [bench.py arguments]   e.g. 2 = never the 128-channel GEMM blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bits = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
from hairfastgan_amd._runtime import lib  # noqa: E402

lib().hf_debug_set_tuning(bits)
import bench  # noqa: E402

bench.main()
