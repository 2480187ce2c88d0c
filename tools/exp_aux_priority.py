import json, os, subprocess, sys
CHILD = r'''
import json, sys
sys.path.insert(0, ".")
import bench, imagestitching_amd as ist
r = bench.file_pipeline_leg(ist)
print(json.dumps({"e2e": r["ms_end_to_end"], "png": r["stages_ms"]["png"], "timed": r["ms_phase_timed_run"]}))
'''
for rnd in range(2):
    for name, env in (("high", {}), ("low", {"IST_TUNING": "1", "IST_AUX_PRIORITY": "0"})):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
