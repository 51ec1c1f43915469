"""conv2 (bf16 x 6 matrix-core kernel) output error against an f64 torch reference (GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from __graft_entry__ import load_package
pkg = load_package(); api, synth = pkg.api, pkg.synth
w = synth.calc_weights()
for name, opt in (("k_conv2_f16x3", 0), ("k_conv2_bf16x6", 1)):
  lcd = api.DeepLCD(w); lcd.set_option(lcd.OPT_CONV2_BF16X6, opt)
  errs = []
  for seed in range(4):
      x = synth._rng(21 + seed).uniform(0, 1, (120, 160)).astype(np.float32)
      o = [0]
      def take(shape):
          n = int(np.prod(shape)); t = torch.from_numpy(w[o[0]:o[0] + n].reshape(shape).copy()).double(); o[0] += n
          return t
      w1, b1, w2, b2 = take((64, 1, 5, 5)), take((64,)), take((128, 64, 4, 4)), take((128,))
      p1 = lcd.debug_forward(x, 1).reshape(31, 41, 64).transpose(2, 0, 1)          # the device's own conv2 INPUT
      a2 = F.relu(F.conv2d(torch.from_numpy(p1.copy())[None].double(), w2, b2, stride=1, padding=2))[0].numpy()
      got = lcd.debug_forward(x, 2).reshape(32, 42, 128).transpose(2, 0, 1)
      errs.append((np.abs(got - a2).max() / np.abs(a2).max(), np.abs(got - a2).mean() / np.abs(a2).mean()))
  print("%s: max-normalised error %.3e, mean relative error %.3e" % ((name,) + tuple(np.max(errs, axis=0))))
