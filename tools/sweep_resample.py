#!/usr/bin/env python3
"""A/B sweep of the resample paths (SAMPLE_LDS tile shape / budget / stages, SWAP_LDS) on the supplementary workloads:
mixed-size strips, the reference's phone-capped plans, scaled EXIF quarter turns.  One process, interleaved rounds, every
variant checked against the first (<= 1 LSB).  Tuning mode: the compile knobs are re-read at every compile.
usage: python tools/sweep_resample.py [rounds] [workload,...] [variant,...]"""
import json
import os
import sys

os.environ["IST_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
UNI = [(4032, 3024)] * 9
KNOBS = ("IST_LDS_TILE_W", "IST_LDS_BUDGET", "IST_LDS_RUN", "IST_NO_LDS", "IST_NO_SORT")
VARIANTS = {
    "auto": {},
    "w256": {"IST_LDS_TILE_W": "256"},
    "w128": {"IST_LDS_TILE_W": "128"},
    "w64": {"IST_LDS_TILE_W": "64"},
    "b16k": {"IST_LDS_BUDGET": "16384"},
    "b20k": {"IST_LDS_BUDGET": "20480"},
    "b32k": {"IST_LDS_BUDGET": "32768"},
    "b40k": {"IST_LDS_BUDGET": "40960"},
    "run1": {"IST_LDS_RUN": "1"},
    "run3": {"IST_LDS_RUN": "3"},
    "run4": {"IST_LDS_RUN": "4"},
    "direct": {"IST_NO_LDS": "1"},
    "nosort": {"IST_NO_SORT": "1"},
}


def workloads():
    def img(w, h, o=1):
        return {"width": w, "height": h, "orientation": o, "opaque": True, "bmpWidth": (h if o >= 5 else w), "bmpHeight": (w if o >= 5 else h)}
    return {
        "mixed_v": ([img(w, h) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
        "mixed_h": ([img(w, h) for w, h in MIXED], "horizontal", {"filter": "bilinear"}),
        "mixed_v_max": ([img(w, h) for w, h in MIXED], "vertical", {"filter": "bilinear", "mode": "max"}),
        "ios_bilinear": ([img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "ios", "superSample": 1}),
        "ios_nearest": ([img(w, h) for w, h in UNI], "vertical", {"filter": "nearest", "platform": "ios", "superSample": 1}),
        "android_bilinear": ([img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "android", "superSample": 1}),
        "orient6_mixed": ([img(w, h, 6) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
        "orient3_mixed": ([img(w, h, 3) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
        "shrink3": ([img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "ios", "superSample": 1, "maxSide": 9072}),   # 1/3 scale
        "shrink3_nearest": ([img(w, h) for w, h in UNI], "vertical", {"filter": "nearest", "platform": "ios", "superSample": 1, "maxSide": 9072}),
    }


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    wl = workloads()
    names = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else list(wl)
    variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["auto", "w256"]
    dev = torch.device("cuda", 0)
    st = ist.Stitcher(0)
    for name in names:
        imgs, direction, opts = wl[name]
        jobs = {}
        for v in variants:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(VARIANTS[v])
            p, jobs[v] = st.compile(imgs, direction, opts)
        for k in KNOBS:
            os.environ.pop(k, None)
        sets = [[torch.randint(0, 256, (i["bmpHeight"], i["bmpWidth"], 4), dtype=torch.uint8, device=dev) for i in imgs] for _ in range(2)]
        for s in sets:
            for t in s:
                t[..., 3] = 255
        outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        ref = None
        for v, job in jobs.items():
            outs[0].fill_(0)
            torch.cuda.synchronize()
            job.launch(sets[0], outs[0])
            torch.cuda.synchronize()
            if ref is None:
                ref = outs[0].clone()
            else:
                d = int((outs[0].to(torch.int16) - ref.to(torch.int16)).abs().max())
                assert d <= 1, (name, v, d)
        res = {v: [] for v in variants}
        for r in range(rounds):
            for v, job in jobs.items():
                for i in range(3):
                    job.launch(sets[i % 2], outs[i % 2])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 30
                e0.record()
                for i in range(n):
                    job.launch(sets[i % 2], outs[i % 2])
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) * 1e3 / n)
        for v, job in jobs.items():
            t = sorted(res[v])
            med = t[len(t) // 2]
            B = job.info["algorithmic_bytes"]
            row = {"workload": name, "variant": v, "canvas": [p.canvas_w, p.canvas_h], "median_us": round(med, 2), "min_us": round(t[0], 2),
                   "algorithmic_bytes": B, "GBs": round(B / med / 1e3, 1), "frac": round(B / med / 1e3 / 8000, 4), "n_tiles": job.info["n_tiles"],
                   "tiles": [job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")]}
            print(json.dumps(row), flush=True)
        del sets, outs, jobs, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
