#!/usr/bin/env python3
"""bench.py on another build of the library (A/B of compiler settings): SR_LIB_PATH=<.so> python tools/bench_with_lib.py <bench.py arguments>.
The product never looks at SR_LIB_PATH; this wrapper points socioreasoner_amd.lib at the file before bench.py imports anything."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from socioreasoner_amd import lib  # noqa: E402

lib.LIB_PATH = os.path.abspath(os.environ["SR_LIB_PATH"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
