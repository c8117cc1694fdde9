import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn as onn
from unsupervised_anomaly_detection_brain_mri_amd import _lib
from tests.gpu_util import dev, ptr, desc, stream
lib = _lib.load()
N, H, Cin, Cout = 2, 16, 32, 64
rng = np.random.default_rng(1)
x = rng.standard_normal((N, H, H, Cin)); w = rng.standard_normal((5, 5, Cin, Cout)) / np.sqrt(25 * Cin); b = rng.standard_normal(Cout)
ref = onn.conv2d_fwd(x, w, b, 2)
d = desc(N, H, H, Cin, H // 2, H // 2, Cout, 5, 2, 1)
xd, wd, bd = dev(x), dev(w), dev(b)
out = torch.full((N, H // 2, H // 2, Cout), -7.0, device='cuda')
_lib.check(lib.uad_op_conv_f(C.byref(d), ptr(xd), None, ptr(wd), ptr(bd), None, None, ptr(out), stream()))
torch.cuda.synchronize()
o = out.cpu().numpy()
print('untouched', (o == -7).mean(), 'nan', np.isnan(o).mean())
err = np.abs(o - ref)
print('max err', np.nanmax(err), 'argmax', np.unravel_index(np.nanargmax(err), err.shape))
print('err by n', np.nanmax(err, axis=(1, 2, 3)))
print('err by oy', np.nanmax(err, axis=(0, 2, 3)))
print('err by ox', np.nanmax(err, axis=(0, 1, 3)))
print('err by co[::8]', np.nanmax(err, axis=(0, 1, 2))[::8])
print(o[0, 0, 0, :4], ref[0, 0, 0, :4])
