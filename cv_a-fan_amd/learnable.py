"""Learnable multi-layer A-FAN: one training iteration of Classification/main_learnable.py:196-255.

Nine feature depths of ResNet-56s are attacked with the same K-step sign-PGD (`attack_algo.PGD`), each adversarial map is
blended back into its clean map with a learnable weight, `clean + w[i] * (adv - clean)` (:226), and the nine mixed tails
plus the clean pass are trained jointly with an L1 penalty on `w` (:240-245); after both optimizers stepped, `w` is
shifted back onto sum(w) = 1 (:254-255, `sum_project`).

The per-element work runs in libafan_hip.so: the PGD steps, the tails (BatchNorm / convolution kernels of resnet_s), the
blend and its d/dw reduction (`afan_mix_w*`), the whole-arena SGD.  The nine-element bookkeeping of `w` (L1 norm, its
own momentum SGD, the projection) stays on torch ops, as in the reference."""
import torch

from .grid_guard import GuardedTrainer

from . import ops
from .arena import ArenaSGD, ParamArena
from .attack_algo import PGD
from . import resnet_s
from .resnet_s import _like_layout

LEARNABLE_IDX = (4, 8, 11, 14, 18, 21, 24, 28, 31)   # main_learnable.py:59


def sum_project(b, K=9):
    """main_learnable.py:369-378."""
    return b - (torch.sum(b, dim=0) - 1) / K


class _MixW(torch.autograd.Function):
    """mixed = clean + w[idx] * (adv - clean); gradient flows to `w` only (clean is detached; nobody reads adv.grad)."""

    @staticmethod
    def forward(ctx, w, idx, clean, adv, out_dtype):
        ctx.save_for_backward(clean, adv)
        ctx.idx, ctx.nw = idx, w.numel()
        return ops.mix_w(clean, adv, w.detach()[idx:idx + 1], out_dtype)

    @staticmethod
    def backward(ctx, g):
        clean, adv = ctx.saved_tensors
        dw = torch.zeros(ctx.nw, dtype=torch.float32, device=clean.device)
        ops.mix_w_backward(_like_layout(g, clean), clean, adv, dw[ctx.idx:ctx.idx + 1])
        return dw, None, None, None, None


class LearnableTrainer(GuardedTrainer):
    """Owns the backbone arena + SGD (sequential_model parameters only, :82-84) and the optimizer of `w` (:86-90)."""

    def __init__(self, model, criterion, *, steps=3, gamma=1.0, eps=2.0, idx_list=LEARNABLE_IDX, layer_number=None,
                 randinit=False, clip=False, lr=0.1, w_lr=0.01, l1_coef=1.0, momentum=0.9, weight_decay=5e-4,
                 use_graph=True, graph_warmup=3):
        self.model, self.criterion = model, criterion
        self.steps, self.gamma, self.eps = steps, gamma, eps
        self.idx_list = tuple(idx_list)
        self.layer_number = layer_number if layer_number is not None else model.layer_number
        if len(self.idx_list) != model.w.numel():
            raise ValueError("one mixing weight per perturbed depth")
        self.randinit, self.clip, self.l1_coef = randinit, clip, l1_coef
        # every parameter except `w`; optimizer state keyed like SGD(model.sequential_model.parameters()) (:82-84)
        self.arena = ParamArena(model, index_root=getattr(model, "sequential_model", None))
        self.optimizer = ArenaSGD(self.arena, lr, momentum, weight_decay)
        self.optimizer_w = torch.optim.SGD([{"params": model.w, "lr": w_lr, "weight_decay": 0}], w_lr,
                                           momentum=momentum, weight_decay=0)
        # hipGraph replay of the whole iteration (~2 500 launches on ResNet-56s), as AfanTrainer does; randinit draws on
        # the host generator every step and therefore stays eager
        self.use_graph = bool(use_graph) and not randinit
        self.graph_warmup = graph_warmup
        self._graph = self._graph_failed = self._static = self._out = self._key = None
        self._eager_steps = 0
        self._guard_init(model, self.arena.param.device)      # a grid barrier that gives up: grid_guard.py

    # `w` and its optimizer are torch's own (nine floats): the device guard does not cover them, the ring keeps their clones
    def _guard_state(self):
        st = self.optimizer_w.state.get(self.model.w, {})
        mb = st.get("momentum_buffer")
        return (self.model.w.detach().clone(), None if mb is None else mb.clone())

    def _guard_restore(self, state):
        w, mb = state
        with torch.no_grad():
            self.model.w.copy_(w)
        st = self.optimizer_w.state.get(self.model.w)
        if st is not None:
            if mb is None:
                st.pop("momentum_buffer", None)
            else:
                st["momentum_buffer"].copy_(mb)

    def _drop_graphs(self):
        self._graph = self._graph_failed = self._static = self._out = self._key = None

    def step(self, inp, target):
        """One iteration; returns device tensors (loss, loss_clean, loss_adv, l1, l2[9,N], linf[9,N], prec1, w).
        `flush_guard()` at every logging interval (grid_guard.py)."""
        return self._guarded((inp, target), self._step_once)

    def _step_once(self, inp, target):
        key = (tuple(inp.shape), inp.dtype, tuple(target.shape))
        if self._graph is not None and self._key == key:
            return self._replay(inp, target)
        if (self.use_graph and self._graph is None and self._graph_failed is None and inp.is_cuda
                and self._eager_steps >= self.graph_warmup and self.model.training and self._graph_safe()):
            try:
                self._capture(inp, target, key)
                return self._replay(inp, target)
            except Exception as e:  # noqa: BLE001 — stay correct: fall back to eager launches, loudly
                import warnings
                self._graph, self._graph_failed = None, e
                warnings.warn(f"hipGraph capture of the learnable A-FAN step failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
        self._eager_steps += 1
        self.optimizer._sync_lr()
        return self._body(inp, target)

    def _graph_safe(self):
        """As AfanTrainer._graph_safe (every configuration runs on the library's own kernels: nothing to exclude)."""
        if resnet_s.vendor_convs(self.model):
            self.use_graph = False
        return self.use_graph

    def _capture(self, inp, target, key):
        dev = inp.device
        self._static = (torch.empty_like(inp), torch.empty_like(target))
        self._static[0].copy_(inp)
        self._static[1].copy_(target)
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with ops.no_gc_during_capture(), torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            out = self._body(self._static[0], self._static[1])
        self._graph, self._out, self._key = g, out, key

    def _replay(self, inp, target):
        self._static[0].copy_(inp, non_blocking=True)
        self._static[1].copy_(target, non_blocking=True)
        self.optimizer._sync_lr()          # lr lives in device memory: the graph reads it, the host only writes it here
        self._graph.replay()
        return {k: v.clone() for k, v in self._out.items() if k != "out_clean"} | {"out_clean": self._out["out_clean"]}

    def _body(self, inp, target):
        m, ln = self.model, self.layer_number
        if inp.is_cuda:
            ops.acc_reset(inp.device)
        clean, adv = [], []
        for num in self.idx_list:
            with torch.no_grad():                            # .detach() at :203 — same values, same BN side effects
                fea = m(inp, end_point=num, start_point=0)
            fea = fea.float() if fea.dtype != torch.float32 else fea
            clean.append(fea)
            adv.append(PGD(fea, self.criterion, y=target, model=m, steps=self.steps, gamma=(self.gamma / 255),
                           start_idx=num, layer_number=ln, eps=(self.eps / 255), randinit=self.randinit,
                           clip=self.clip))
        dtype = getattr(m, "compute_dtype", torch.float32)
        outs, l2s, linfs = [], [], []
        for i, num in enumerate(self.idx_list):
            l2, linf = ops.perturb_norms(adv[i].detach(), clean[i])      # :219-220, on the device
            l2s.append(l2)
            linfs.append(linf)
            mixed = _MixW.apply(m.w, i, clean[i], adv[i].detach(), dtype)
            outs.append(m(mixed, end_point=ln, start_point=num))
        out_clean = m(inp, end_point=ln, start_point=0)
        loss_adv = 0
        for o in outs:
            loss_adv = loss_adv + self.criterion(o.float(), target)
        loss_clean = self.criterion(out_clean.float(), target)
        l1 = torch.norm(m.w, p=1)
        loss = (loss_clean + loss_adv / len(self.idx_list)) / 2 + l1 * self.l1_coef
        self.optimizer.zero_grad()
        self.optimizer_w.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.optimizer_w.step()
        with torch.no_grad():
            m.w.data.copy_(sum_project(m.w.data, K=len(self.idx_list)))   # in place (same values): graph-capturable
        prec1 = (out_clean.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
        return {"loss": loss.detach(), "loss_clean": loss_clean.detach(), "loss_adv": loss_adv.detach(),
                "l1": l1.detach(), "l2": torch.stack(l2s), "linf": torch.stack(linfs), "prec1": prec1,
                "w": m.w.detach().clone(), "out_clean": out_clean.detach()}
