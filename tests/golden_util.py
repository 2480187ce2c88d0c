"""Turn a captured reference call trace (tests/golden/plan_goldens.json) into comparable pieces, and generate the
call sequence a plan implies (test-side restatement of utils/canvas.js:153-202 + index.js:1391-1428,1559-1579)."""
import math

HALF_PI = 0.5 * math.pi


def case_inputs(case):
    inp, lim = case["input"], case["limits"]
    descs = [{"width": im["w"], "height": im["h"], "orientation": im.get("orientation", 1), "file_size": im.get("fileSize", 0)}
             for im in inp["images"]]
    return inp, lim, descs


def strip(calls):
    """Drop the bookkeeping keys so traces compare as [op, args...] lists."""
    out = []
    for e in calls:
        if e["op"] == "fillStyle":
            out.append(["fillStyle", e["v"]])
        elif e["op"] == "drawImage":
            out.append(["drawImage", e["img"], e["bmp"], e["a"]])
        elif e["op"] == "export":
            out.append(["export", e["a"], e["fileType"], e["quality"]])
        elif "a" in e:
            out.append([e["op"], e["a"]])
        else:
            out.append([e["op"]])
    return out


def expected_calls(plan, rects, descs):
    """plan: dict with canvas_w, canvas_h, super_sample, big_task; rects: list of dicts."""
    cw, ch = plan["canvas_w"], plan["canvas_h"]
    calls = [["createOffscreenCanvas", [cw, ch]], ["fillStyle", "#ffffff"], ["fillRect", [0, 0, cw, ch]]]
    if plan["super_sample"] != 1:
        calls.append(["scale", [plan["super_sample"], plan["super_sample"]]])
    for r in rects:
        d = descs[r["image"]]
        bw, bh = d.get("bmp_w") or d["width"], d.get("bmp_h") or d["height"]
        dx, dy, dw, dh, o = r["dx"], r["dy"], r["dw"], r["dh"], r["orientation"]
        calls.append(["save"])
        src = [0, 0, bw, bh]
        if not o or o == 1 or o > 8:
            calls.append(["drawImage", r["image"], [bw, bh], src + [dx, dy, dw, dh]])
        else:
            pre = {
                2: [["translate", [dx + dw, dy]], ["scale", [-1, 1]]],
                3: [["translate", [dx + dw, dy + dh]], ["rotate", [math.pi]]],
                4: [["translate", [dx, dy + dh]], ["scale", [1, -1]]],
                5: [["translate", [dx, dy]], ["rotate", [HALF_PI]], ["scale", [1, -1]]],
                6: [["translate", [dx + dw, dy]], ["rotate", [HALF_PI]]],
                7: [["translate", [dx + dw, dy]], ["rotate", [HALF_PI]], ["scale", [-1, 1]]],
                8: [["translate", [dx, dy + dh]], ["rotate", [-HALF_PI]]],
            }[o]
            calls += pre
            dst = [0, 0, dw, dh] if o <= 4 else [0, 0, dh, dw]
            calls.append(["drawImage", r["image"], [bw, bh], src + dst])
        calls.append(["restore"])
        if plan["big_task"]:
            calls.append(["getImageData", [0, 0, 1, 1]])
    calls.append(["export", [0, 0, cw, ch, cw, ch], "png", 1])
    return calls


class Ctm:
    """Canvas CTM replay with the quarter-turn rule of DESIGN.md (exact cos/sin for multiples of pi/2)."""

    def __init__(self):
        self.m = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
        self.stack = []

    def save(self):
        self.stack.append(list(self.m))

    def restore(self):
        self.m = self.stack.pop()

    def translate(self, x, y):
        a, b, c, d, e, f = self.m
        self.m = [a, b, c, d, a * x + c * y + e, b * x + d * y + f]

    def scale(self, x, y):
        a, b, c, d, e, f = self.m
        self.m = [a * x, b * x, c * y, d * y, e, f]

    def rotate(self, r):
        q = r / 1.5707963267948966
        qr = math.floor(q + 0.5)
        if abs(q - qr) < 1e-9:
            k = int(qr) % 4
            co, si = [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)][k]
        else:
            co, si = math.cos(r), math.sin(r)
        a, b, c, d, e, f = self.m
        self.m = [a * co + c * si, b * co + d * si, c * co - a * si, d * co - b * si, e, f]


def replay_draws(calls):
    """Replay a captured trace; returns [(image, m[6], s[4], d[4])] for every drawImage, plus the fill list."""
    ctm = Ctm()
    draws, fills = [], []
    for e in calls:
        op = e["op"]
        if op in ("save", "restore"):
            getattr(ctm, op)()
        elif op in ("translate", "scale", "rotate"):
            getattr(ctm, op)(*e["a"])
        elif op == "fillRect":
            fills.append((list(ctm.m), list(e["a"])))
        elif op == "drawImage":
            a = e["a"]
            draws.append((e["img"], list(ctm.m), [float(v) for v in a[:4]], [float(v) for v in a[4:]]))
    return draws, fills
