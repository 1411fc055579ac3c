"""Generates tests/golden/*.npz FROM THE REFERENCE (run in the build container only):

    python tests/golden/make_golden.py

Imports the reference's own model files from /root/reference via oracle/refload.py, fills the
weights with the name-keyed deterministic fill (oracle.deterministic_state -- a formula, so the
20 M weights need not be stored), runs one G-step, one D-step (each from the fresh state) and an
eval forward, and records inputs + expected outputs.  The files are data only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mixstage_oracle as O   # noqa: E402
from oracle import refload                # noqa: E402

CONFIGS = {
    # name: (B, T, M, S, dtype, data seed)
    'c1_fp32': (4, 64, 1, 2, torch.float32, 1234),     # BASELINE configs[0] (S=2: S=1 crashes, SURVEY s.0 item 4)
    'c1_fp64': (4, 64, 1, 2, torch.float64, 1234),
    # The data seeds of the two multi-speaker fixtures are SCREENED (tests/golden/screen_seed.py, seeds 2000-2159), by two criteria.
    # (1) The discriminator's gradients are not a smooth function of its input where a pre-activation sits on a LeakyReLU kink, and
    # with seed 1234 the smallest |pre-activation| in D was 6.7e-6 (D-step) / 2.3e-6 (G-step) -- inside what two fp32 summation
    # orders differ by.  The five seeds with the largest margin were kept (recorded below as `d_margin`); 1e-4 was not reached by any
    # (D holds ~6e4 pre-activations per step with density ~1 around zero: the expected minimum is ~1e-5).  (2) The generator has
    # ~1e7 activations, so SOME sit within fp32 rounding of a kink for every seed; what can be screened is how much that moves the
    # gradients: among the five, the seed whose fp32 and fp64 reference G-steps agree best in every parameter's gradient norm
    # (`screen_seed.py robust`: 1.2e-4 / 2.3e-4 for the seeds below; 7.2e-3 for the margin-best c3r seed 2065, which two fp32
    # implementations cannot both match at the 2e-3 norm bar).
    'c2r_fp32': (4, 64, 4, 4, torch.float32, 2069),    # BASELINE configs[1] at reduced batch; margins 2.1e-5 (D-step), 3.7e-5 (G-step)
    'c3r_fp32': (2, 64, 8, 8, torch.float32, 2058),    # headline M=S=8 at reduced batch; margins 4.3e-5, 5.9e-5
}
BN_PROBES = ['G.decoder.0.norm', 'G.audio_encoder.conv.7.norm', 'D.conv3.norm']


def grad_probe(model):
  out = {}
  for n, p in model.named_parameters():
    if p.grad is None:
      continue
    g = p.grad.detach().double().reshape(-1)
    stride = max(1, g.numel() // 16)
    out['gnorm/' + n] = np.float64(g.norm().item())
    out['gsamp/' + n] = g[::stride][:16].numpy()
  return out


def param_probe(model, which):
  out = {}
  mod = model.G if which == 'G' else model.D
  for n, p in mod.named_parameters():
    v = p.detach().double().reshape(-1)
    out['psum/%s.%s' % (which, n)] = np.float64(v.sum().item())
    out['pnorm/%s.%s' % (which, n)] = np.float64(v.norm().item())
  return out


def run(name, B, T, M, S, dtype, seed=1234):
  audio, pose, labels, style = O.synthetic_batch(B, T=T, M=M, S=S, dtype=dtype, seed=seed)
  rec = dict(audio=audio.numpy(), pose=pose.numpy(), labels=labels.numpy(), style=style.numpy(),
             meta=np.array([B, T, M, S], dtype=np.int64), data_seed=np.int64(seed))
  for kind in ('G', 'D'):
    ref = refload.build_ref_gan(M=M, S=S, T=T, dtype=None)
    ref.load_state_dict(O.deterministic_state(ref.state_dict()))
    ref.to(dtype)
    caps = {}
    dcalls = []

    def pse_hook(m, i, o):          # hooks must return None (a value would replace the output)
      caps.setdefault('pse', o.detach().clone())

    def d_hook(m, i, o):
      dcalls.append(o[0].detach().clone())

    dmin = []
    hk = [m.register_forward_pre_hook(lambda mod, inp: dmin.append(float(inp[0].detach().abs().min())))
          for m in ref.D.modules() if isinstance(m, torch.nn.LeakyReLU)]
    h1 = ref.G.pose_style_encoder.register_forward_hook(pse_hook)
    h2 = ref.D.register_forward_hook(d_hook)
    og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
    od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
    torch.manual_seed(7)
    # forward/backward/clip/Adam through the trainer-step contract (model-agnostic driver)
    fake, losses, gnorm = O.oracle_train_step(ref, og, od, audio, pose, labels, style, kind, T=T)
    h1.remove(); h2.remove()
    for h in hk:
      h.remove()
    k = kind + '/'
    rec[k + 'd_margin'] = np.float64(min(dmin))      # smallest |input| of any LeakyReLU of the discriminator in this step
    rec[k + 'pose'] = fake.numpy()
    rec[k + 'losses'] = np.array(losses, dtype=np.float64)
    rec[k + 'total_grad_norm'] = np.float64(gnorm)
    rec[k + 'labels_cap_soft'] = ref.G.labels_cap_soft.detach().numpy()
    rec[k + 'dscores'] = np.stack([d.numpy() for d in dcalls])
    if 'pse' in caps:
      sc = caps['pse'].double()
      top2 = sc.topk(2, dim=-1).values
      rec[k + 'pse_score'] = sc.numpy()
      rec[k + 'pse_argmax'] = sc.argmax(-1).numpy()
      rec[k + 'pse_margin'] = (top2[:, 0] - top2[:, 1]).numpy()
    # note: clip_grad_norm_ rescaled the grads in place before we probe them
    for kk, v in grad_probe(ref).items():
      rec[k + kk] = v
    for kk, v in param_probe(ref, kind).items():
      rec[k + kk] = v
    sd = ref.state_dict()
    for b in BN_PROBES:
      rec[k + 'bn/' + b + '.running_mean'] = sd[b + '.running_mean'].double().numpy()
      rec[k + 'bn/' + b + '.running_var'] = sd[b + '.running_var'].double().numpy()
  # eval forward (running statistics, no update)
  ref = refload.build_ref_gan(M=M, S=S, T=T, dtype=None)
  ref.load_state_dict(O.deterministic_state(ref.state_dict()))
  ref.to(dtype).eval()
  with torch.no_grad():
    fake, losses, _ = ref([audio, labels], pose, **O.model_kwargs(style, T))
  rec['E/pose'] = fake.numpy()
  rec['E/losses'] = np.array([float(l) for l in losses], dtype=np.float64)
  # inference branch: sample_flag=1 takes the style ids from kwargs (line 168-174)
  kw = O.model_kwargs(style, T); kw['sample_flag'] = 1
  with torch.no_grad():
    fake, losses, _ = ref([audio, labels], pose, **kw)
  rec['S/pose'] = fake.numpy()
  np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '.npz'), **rec)
  print(name, 'ok', {k: (v.shape if hasattr(v, 'shape') else v) for k, v in list(rec.items())[:6]})


def run_n1n3():
  """Pre-step (N1) and step-metric (N3) vectors from the reference's own transform.py / metrics.py (refload stubs the
  dataset stack they import): KMeans labels with the job scripts' features [pose, velocity, speed], ZNorm, L1 / VelL1 / PCK.
  The joint removal in front of KMeans is the oracle's (pycasper.remove_slices is not in the reference tree)."""
  import types
  from oracle import prestep_oracle as PO
  tm = refload.load_transform_and_metrics()
  T, Mx = tm.transform, tm.metrics
  rng = np.random.default_rng(2024)
  B, Tn, P, M, mask = 6, 64, 104, 8, [0, 7, 8, 9]
  pose = (rng.standard_normal((B, Tn, P)) * 40 + 100).astype(np.float32)
  pose[:, 1:] = pose[:, :1] + np.cumsum(rng.standard_normal((B, Tn - 1, P)).astype(np.float32), axis=1)
  PK = P - 2 * len(mask)
  centers = np.concatenate([rng.standard_normal((M, PK)) * 40 + 100, rng.standard_normal((M, PK)),
                            np.abs(rng.standard_normal((M, PK // 2))) * 1.5], axis=1)
  mean, var = rng.standard_normal(P) * 10 + 100, rng.random(P) * 50 + 1
  var[5], var[11] = 0.0, -1e-9
  kept = torch.from_numpy(PO.remove_joints(pose, mask))
  rec = dict(pose=pose, centers=centers, mean=mean, var=var, mask=np.array(mask))
  for tag, feats, width in (('pvs', ['pose', 'velocity', 'speed'], 2 * PK + PK // 2), ('pv', ['pose', 'velocity'], 2 * PK)):
    km = types.SimpleNamespace(feats=feats, centers=torch.from_numpy(centers[:, :width].copy()))
    km.get_feats = lambda x, km=km: T.KMeans.get_feats(km, x)
    rec['labels_' + tag] = T.KMeans.predict(km, kept).numpy()
  rec['znorm'] = T.ZNorm.znorm(None, torch.from_numpy(pose).double(), [torch.from_numpy(mean), torch.from_numpy(var)]).numpy()
  y = rng.standard_normal((B, Tn, P)); gt = y + 0.3 * rng.standard_normal((B, Tn, P))
  l1, vl, pck = Mx.L1(), Mx.VelL1(), Mx.PCK(alphas=[0.1, 0.2], num_joints=52)
  l1(torch.from_numpy(y), torch.from_numpy(gt), mask_idx=mask)
  vl(torch.from_numpy(y), torch.from_numpy(gt), mask_idx=mask)
  pck(torch.from_numpy(y).view(-1, 2, 52), torch.from_numpy(gt).view(-1, 2, 52), mask_idx=mask)
  pa = pck.get_averages('t')
  rec.update(m_y=y, m_gt=gt, L1=np.float64(l1.get_averages('t')['t_L1']), VelL1=np.float64(vl.get_averages('t')['t_VelL1']),
             pck=np.array([[pa['t_pck_%s_%d' % (a, j)] for j in range(52)] for a in (0.1, 0.2)]),
             pck_mean=np.array([pa['t_pck_0.1'], pa['t_pck_0.2']]))
  np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'n1n3.npz'), **rec)
  print('n1n3 ok', {k: getattr(v, 'shape', v) for k, v in rec.items()})


def run_evalacc():
  """FID / W1 (N3) from the reference's own metrics.py over three batches fed as calculate_metrics feeds them (trainer.py:884-896):
  inputs, the running histograms / Gram matrices and the final numbers."""
  from oracle import metrics_oracle as MO
  Mx = refload.load_transform_and_metrics().metrics
  mask = [0, 7, 8, 9]
  kept = [j for j in range(52) if j not in mask]
  rng = np.random.default_rng(29)
  mean, var = rng.standard_normal(104) * 20 + 150, rng.random(104) * 400 + 25
  std = var ** 0.5
  fid, w1 = Mx.FID(), Mx.W1()
  ys, gts = [], []
  B, Tn = 5, 24
  for step in range(3):
    gt = rng.standard_normal((B, 1, 104)) * 0.5 + np.cumsum(rng.standard_normal((B, Tn, 104)) * 0.08, axis=1)
    y_kept = (gt.reshape(B, Tn, 2, 52)[..., kept] + 0.15 * rng.standard_normal((B, Tn, 2, 48))).reshape(B, Tn, 96)
    y_kept, gt = y_kept.astype(np.float32), gt.astype(np.float32)
    ys.append(y_kept); gts.append(gt)
    y_full = MO.reinsert_joints(y_kept.astype(np.float64), gt.astype(np.float64), mask)
    fid(torch.from_numpy(y_full), torch.from_numpy(gt.astype(np.float64)), mask_idx=mask)
    w1(torch.from_numpy((y_full * std + mean).reshape(B, Tn, 2, 52)),
       torch.from_numpy((gt.astype(np.float64) * std + mean).reshape(B, Tn, 2, 52)), mask_idx=mask)
  fa, wa = fid.get_averages('t'), w1.get_averages('t')
  rec = dict(y=np.stack(ys), gt=np.stack(gts), mean=mean, var=var, mask=np.array(mask),
             FID=np.float64(fa['t_FID']), W1_vel=np.float64(wa['t_W1_vel']), W1_acc=np.float64(wa['t_W1_acc']),
             hist=np.stack([w1.y_vel_meter.sum, w1.y_acc_meter.sum, w1.gt_vel_meter.sum, w1.gt_acc_meter.sum]).astype(np.int64),
             y_sum=fid.y_sum_meter.sum.numpy(), gt_sum=fid.gt_sum_meter.sum.numpy(),
             y_square=fid.y_square_meter.sum.numpy(), gt_square=fid.gt_square_meter.sum.numpy())
  np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'evalacc.npz'), **rec)
  print('evalacc ok', {k: getattr(v, 'shape', v) for k, v in rec.items()})


if __name__ == '__main__':
  assert refload.available(), 'needs /root/reference'
  if len(sys.argv) > 1 and sys.argv[1] == 'n1n3':
    run_n1n3()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'evalacc':
    run_evalacc()
    sys.exit(0)
  only = sys.argv[1:]
  for name, cfg in CONFIGS.items():
    if not only or name in only:
      run(name, *cfg)
  if not only:
    run_n1n3()
    run_evalacc()
