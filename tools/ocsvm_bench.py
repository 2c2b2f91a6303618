#!/usr/bin/env python3
"""seg_fuser.fit at the size of one MVTec-3D class (244 train images x 50 176 pixels x 2 score maps = 12.2 M rows): scikit-learn on
the host (what the reference runs) against cmdiad_ocsvm_fit on the device; both must return the same model."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sklearn import linear_model
from cmdiad_amd.ocsvm import DeviceSGDOneClassSVM
n = int(os.environ.get("OCSVM_N", 244 * 50176))
g = np.random.default_rng(0)
X = np.abs(g.normal(1.0, 0.25, size=(n, 2))).astype(np.float32)
t = time.time(); ref = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(X); th = time.time() - t
Xd = torch.from_numpy(X).cuda()
DeviceSGDOneClassSVM(random_state=42, nu=0.5, max_iter=1).fit(Xd[:100000])   # warm-up (module load)
torch.cuda.synchronize(); t = time.time(); dev = DeviceSGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(Xd); torch.cuda.synchronize(); td = time.time() - t
print(f"n = {n}: scikit-learn host fit {th:.2f} s ({ref.n_iter_} epochs, {th / ref.n_iter_:.3f} s/epoch); device fit {td:.2f} s ({dev.n_iter_} epochs, "
      f"{td / dev.n_iter_:.3f} s/epoch); identical model: {np.array_equal(ref.coef_, dev.coef_) and np.array_equal(ref.offset_, dev.offset_)}", flush=True)
