import io, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from PIL import Image
import imagestitching_amd as ist
from tests import util as U
a = U.smooth_image(1, 48, 64)[..., :3]
b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2)
data = b.getvalue()
got = ist.decode_image(data).astype(int); ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")).astype(int)
d = np.abs(got - ref)
print("per channel max", d.reshape(-1, 4).max(0))
print("rows wrong", np.nonzero(d.max((1, 2)))[0][:20], "cols wrong", np.nonzero(d.max((0, 2)))[0][:40])
print("got[0,:8]", got[0, :8].tolist()); print("ref[0,:8]", ref[0, :8].tolist())
print("got[1,:8]", got[1, :8].tolist()); print("ref[1,:8]", ref[1, :8].tolist())
