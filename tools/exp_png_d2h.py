#!/usr/bin/env python3
"""What do the slabs' device-to-host copies cost the compressing kernel beside them?  The PNG stage of the phase-timed file
pipeline with and without the copies (IST_TUNING=1 IST_PNG_SKIP_D2H=1: the file is then not delivered - timing only)."""
import json, os, subprocess, sys
CHILD = r'''
import json, sys
sys.path.insert(0, ".")
import bench, imagestitching_amd as ist
r = bench.file_pipeline_leg(ist)
print(json.dumps({"png_stage_ms": r["stages_ms"]["png"], "e2e": r["ms_end_to_end"]}))
'''
for rnd in range(2):
    for name, env in (("runtime_copy", {}), ("no_copies", {"IST_TUNING": "1", "IST_PNG_SKIP_D2H": "1"}),
                      ("kernel_32", {"IST_TUNING": "1", "IST_PNG_COPY_GRID": "32"}), ("kernel_64", {"IST_TUNING": "1", "IST_PNG_COPY_GRID": "64"}),
                      ("kernel_128", {"IST_TUNING": "1", "IST_PNG_COPY_GRID": "128"}), ("kernel_256", {"IST_TUNING": "1", "IST_PNG_COPY_GRID": "256"}),
                      ("kernel_1024", {"IST_TUNING": "1", "IST_PNG_COPY_GRID": "1024"})):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
