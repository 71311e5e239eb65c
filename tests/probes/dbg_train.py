import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import conftest, helpers, torch, numpy as np
import vsr_oracle as vo
from vsrcap import synth
from test_gpu_train import _oracle_grads, _device_grads
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = dict(V=50, B=4, R0=10, R=10, D=512, L=5, T=T, E=64, H=64, A=32)
w = synth.make_weights(cfg["V"], cfg["D"], cfg["E"], cfg["H"], cfg["A"], seed=0, gains={k: 1.0 for k in synth.DEFAULT_GAINS})
det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, 3)
if T == 1:
    caps = torch.cat([caps, caps], 1)[:, :1]
m = helpers.build_model(cfg, w, "cuda")
def loss_fn(out, gate, caps, gts):
    V = out.shape[-1]
    return (-out.gather(2, caps[:, :, None].to(out.device)).mean()) + 4 * torch.nn.functional.nll_loss(gate.reshape(-1, 2), gts.reshape(-1).long().to(out.device), ignore_index=-1)
m.train(); m.zero_grad()
out, gate = m((det.cuda(),), (caps.cuda(), ctrl_seq.cuda()))
loss = loss_fn(out, gate, caps, gts); loss.backward()
got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
o = vo.Oracle(w, T, 2, as_written=True)
for k in o.p: o.p[k].requires_grad_(True)
oo, og = o.forward(det, caps, ctrl_seq)
ol = loss_fn(oo, og, caps, gts); ol.backward()
print("loss", loss.item(), ol.item())
for k in o.p:
    gg, r = got[k].double(), o.p[k].grad.double()
    print("%-28s scale %.3e maxerr %.3e rel %.2e" % (k, r.abs().max(), (gg-r).abs().max(), (gg-r).abs().max()/(r.abs().max()+1e-30)))
H = cfg["H"]
for name in ("lstm_cell_2.bias_ih", "lstm_cell_1.bias_ih"):
    gg, r = got[name].double(), o.p[name].grad.double()
    for q, nm in enumerate("ifgo"):
        a, b = gg[q*H:(q+1)*H], r[q*H:(q+1)*H]
        print(name, nm, "scale %.3e maxerr %.3e" % (b.abs().max(), (a-b).abs().max()))
# ---- localise: dh2 at T == 1 is dlogits . W_out
eng = m._eng
B, V = cfg["B"], cfg["V"]
Vp = (V + 3) // 4 * 4
dl = eng.debug_buffer("dlogits", (T * B, Vp), "cuda").cpu()[:, :V].double()
dh2 = eng.debug_buffer("dh2_voc", (T * B, H), "cuda").cpu().double()
wt = eng.debug_buffer("wT_out", (H, Vp), "cuda").cpu().double()
W = torch.from_numpy(w["out_fc.weight"]).double()
print("wT_out err", (wt[:, :V] - W.t()).abs().max().item(), "pad", wt[:, V:].abs().max().item() if Vp > V else 0)
print("dh2_voc err", (dh2 - dl @ W).abs().max().item(), "scale", (dl @ W).abs().max().item())
g2 = eng.debug_buffer("gates2", (T * B, 4 * H), "cuda").cpu().double()
c2 = eng.debug_buffer("c2s", ((T + 1) * B, H), "cuda").cpu().double()[B:]
dp2 = eng.debug_buffer("dpre2", (T * B, 4 * H), "cuda").cpu().double()
dh = dl @ W
i, f, g, og = g2[:, :H], g2[:, H:2*H], g2[:, 2*H:3*H], g2[:, 3*H:]
tc = torch.tanh(c2)
dc = dh * og * (1 - tc * tc)
exp = torch.cat([dc * g * i * (1 - i), torch.zeros_like(dc), dc * i * (1 - g * g), dh * tc * og * (1 - og)], 1)
print("dpre2 err vs formula", (dp2 - exp).abs().max().item(), "scale", exp.abs().max().item())
print("bias grad oracle vs formula colsum", (o.p["lstm_cell_2.bias_ih"].grad.double() - exp.sum(0)).abs().max().item())
