import sys, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, "tests/golden")
import numpy as np, torch, torch.nn.functional as F
import inputs as gi, oracle
from derl_amd.cnn_engine import CnnEngine
from derl_amd import ops
DEV = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 130
rs = np.random.RandomState(batch)
A = 6
weights = gi.nature_cnn_weights(A, 31)
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pool_all = gi.frames(batch + extra, 1000 + batch)
idx_np = rs.permutation(batch + extra)[:batch].astype(np.int32) if extra else None
pool = pool_all[idx_np] if extra else pool_all
data = dict(observations=pool, actions=rs.randint(0, A, batch).astype(np.int64),
            log_prob=(rs.standard_normal(batch) * 0.1 - 1.7).astype(np.float32),
            advantages=rs.standard_normal(batch).astype(np.float32),
            values=rs.standard_normal((batch, 1)).astype(np.float32) * 0.2,
            value_targets=rs.standard_normal((batch, 1)).astype(np.float32))
eng = CnnEngine(A, max_batch=256, device=DEV); eng.load_state_dict(weights)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
obs = t(pool_all)
sidx = t(idx_np) if extra else None
head = eng.forward(obs, sidx); eng._ensure_backward()
dhead = eng.dhead[:batch*32].view(batch, 32)
loss = ops.categorical_loss(head, t(data["actions"]), t(data["log_prob"]), t(data["advantages"]),
    t(data["values"].reshape(-1)), t(data["value_targets"].reshape(-1)), A, 0, 0.1, 0.25, 0.01, dhead)
eng.backward(obs, sidx); torch.cuda.synchronize()
# oracle with activation grads
P = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in weights.items()}
x = torch.from_numpy(pool).permute(0,3,1,2).float()/255
acts = []
for i, s in enumerate((4,2,1)):
  x = F.relu(F.conv2d(x, P[f"base.conv-{i}.weight"], P[f"base.conv-{i}.bias"], stride=s)); x.retain_grad(); acts.append(x)
hid = F.linear(torch.flatten(x,1), P["base.linear.weight"], P["base.linear.bias"]); hid.retain_grad()
logits = F.linear(hid, P["output_layers.0.weight"], P["output_layers.0.bias"]); logits.retain_grad()
values = F.linear(hid, P["output_layers.1.weight"], P["output_layers.1.bias"]); values.retain_grad()
lp, ent, _ = oracle.categorical_log_prob_entropy(logits, data["actions"])
terms = oracle.ppo_loss_terms(lp, ent, values, data["log_prob"], data["advantages"], data["values"], data["value_targets"], 0.1, 0.25, 0.01)
terms["loss"].backward()
print("loss", loss[0].item(), terms["loss"].item())
def cmp(name, got, ref):
  d = np.abs(got - ref); 
  bad = np.argwhere(d > 1e-5 + 1e-4*np.abs(ref))
  print(f"{name}: max|ref|={np.abs(ref).max():.3e} maxdiff={d.max():.3e} nbad={len(bad)}/{ref.size}", "first bad", bad[:3].tolist() if len(bad) else "")
cmp("dhead_logits", dhead[:, :A].cpu().numpy(), logits.grad.numpy())
cmp("dhead_value", dhead[:, A:A+1].cpu().numpy(), values.grad.numpy())
cmp("dhid", eng.dhid[:batch*512].view(batch,512).cpu().numpy(), hid.grad.numpy())
# relu-masked grads: engine stores grad wrt pre-activation = grad_post * (y>0)
for name, a in zip(("dy2","dy1","dy0"), (acts[2], acts[1], acts[0])):
  ref = (a.grad * (a > 0)).permute(0,2,3,1).contiguous().numpy()
  got = getattr(eng, name)[:ref.size].cpu().numpy().reshape(ref.shape)
  cmp(name, got, ref)
views = eng.named_views(eng.grads)
for k in weights:
  cmp(k, views[k].cpu().numpy(), P[k].grad.numpy())
