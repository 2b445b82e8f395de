"""Learned robot-to-point distance network -- mirrors ``mlp_learn/sdf/robot_sdf.py``
(class RobotSdfCollisionNet, lines 13-166) and the NeRF-encoded MLP of
``network_macros_mod.py:96-146``.  Weights are held as plain fp32 arrays; every evaluation
goes through the HIP kernels (omds_mlp_forward_vjp)."""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine


def linear_plan(in_features, out_channels, layers, skips):
    """The flat sequence of Linear layers MLPRegression builds from (mlp_layers, skips), network_macros_mod.py:113-133:
    the layer list is cut at the skip positions into modules; the last layer of every module but the final one is
    narrowed by the encoded input width (the concatenation restores it), and the first entry of every later module only
    names the width of that concatenated vector.  Returns ([(in, out)], state-dict key stems, skip_after) with
    skip_after = indices of the Linear layers followed by a concatenation."""
    layers = [int(v) for v in layers]
    cuts = [0] + [int(s) for s in skips] + [len(layers)]
    groups = [layers[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    for g in groups[:-1]:
        g[-1] -= in_features
    groups[0] = [in_features] + groups[0]
    groups[-1] = groups[-1] + [out_channels]
    shapes, stems, skip_after = [], [], []
    for gi, g in enumerate(groups):
        for i in range(1, len(g)):
            shapes.append((g[i - 1], g[i]))
            stems.append(f"layers.{gi}.{i - 1}.0")
        if gi < len(groups) - 1:
            skip_after.append(len(shapes) - 1)
    return shapes, stems, skip_after


class _WeightBag:
    """Stands in for ``nn_model.model``: holds W[i] ([out, in] like nn.Linear) and b[i]."""

    def __init__(self, in_channels, out_channels, layers, skips=()):
        shapes, self.stems, self.skip_after = linear_plan(3 * in_channels, out_channels, layers, skips)
        self.W = [np.zeros((o, i), np.float32) for i, o in shapes]
        self.b = [np.zeros(o, np.float32) for _, o in shapes]
        self.act = "relu"

    def to(self, *a, **k):
        return self

    def eval(self):
        return self


class RobotSdfCollisionNet:
    def __init__(self, in_channels, out_channels, skips, layers):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.model = _WeightBag(in_channels, out_channels, layers, skips)
        self.model_jit = self          # drivers call nn_model.model_jit.forward(x)
        self.order = list(range(out_channels))
        self.norm_dict = None
        self._engine = None
        self._engine_cap = 0
        self.device = 0

    def set_link_order(self, order):
        self.order = order

    def load_weights(self, f_name, tensor_args=None):
        """Accepts the reference's ``.pt`` checkpoints (state dict keys layers.0.<i>.0.weight/bias,
        robot_sdf.py:39-41) or this repo's ``.npz`` exports (W<i>, b<i>)."""
        if str(f_name).endswith(".npz"):
            z = np.load(f_name)
            n = len([k for k in z.files if k.startswith("W")])
            W = [z[f"W{i}"].astype(np.float32) for i in range(n)]
            b = [z[f"b{i}"].astype(np.float32) for i in range(n)]
            if "act" in z.files:
                self.model.act = str(z["act"])
        else:
            chk = torch.load(f_name, map_location=torch.device("cpu"), weights_only=False)
            sd = chk["model_state_dict"]
            self.norm_dict = chk.get("norm")
            if self.norm_dict is not None:          # robot_sdf.py:41-44 (.to(**tensor_args): the Franka checkpoint stores fp16)
                self.norm_dict = {k: {kk: torch.as_tensor(vv).to(torch.float32) for kk, vv in v.items()} for k, v in self.norm_dict.items()}
            missing = [st for st in self.model.stems if st + ".weight" not in sd]
            if missing:
                raise ValueError(f"checkpoint has no {missing[0]}.weight: it was not saved from a network with these layers / skips")
            W = [sd[st + ".weight"].numpy().astype(np.float32) for st in self.model.stems]
            b = [sd[st + ".bias"].numpy().astype(np.float32) for st in self.model.stems]
        if [w.shape for w in W] != [w.shape for w in self.model.W]:
            raise ValueError(f"checkpoint shapes {[w.shape for w in W]} do not match the declared network "
                             f"{[w.shape for w in self.model.W]}")
        self.model.W, self.model.b = W, b
        self.tensor_args = tensor_args
        self._engine = None
        print("Weights loaded!")

    # -- no-ops kept for call compatibility (robot_sdf.py:112-115,164-166; frankaPlanner.py:48-51)
    def update_aot_lambda(self):
        return 0

    def allocate_gradients(self, N, tensor_args=None):
        self.maxInputSize = N

    def _eng(self, batch):
        if self._engine is None or batch > self._engine_cap:
            if self._engine is not None:
                self._engine.close()
            cap = max(256, int(batch))
            n = self.in_channels - 3
            self._engine = Engine(n, cap, 1, 1, 1, device=self.device)
            self._engine.set_mlp(self.model.W, self.model.b, self.model.act, skip_after=self.model.skip_after)
            self._engine_cap = cap
        return self._engine

    def forward(self, x):
        """MLPRegression.forward (network_macros_mod.py:137-146): raw outputs [B, C]."""
        x = np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float32)
        y, _, _ = self._eng(x.shape[0]).mlp_forward_vjp(x)
        return torch.from_numpy(y)

    __call__ = forward

    def compute_signed_distance(self, q):
        """robot_sdf.py:52-66; the reference scales through ``norm_dict`` here (checkpoints carry one)."""
        if self.norm_dict is not None:
            y = self._unscale(self.forward(self._scale(q)))
        else:
            y = self.forward(q)
        return y[:, self.order]

    # -- checkpoint normalisation (network_macros_mod.py:170-200) -------------------------------------------
    def _scale(self, q):
        q = torch.as_tensor(q, dtype=torch.float32)
        s = torch.div(q - self.norm_dict["x"]["mean"], self.norm_dict["x"]["std"])
        s[s != s] = 0.0
        return s

    def _unscale(self, y):
        return torch.mul(y, self.norm_dict["y"]["std"]) + self.norm_dict["y"]["mean"]

    @staticmethod
    def _rows(q):
        return np.asarray(q.detach().cpu() if isinstance(q, torch.Tensor) else q, dtype=np.float32)

    def compute_signed_distance_wgrad(self, q, idx="all"):
        """(dist [B,C] in ``self.order``, grads, minidx) -- robot_sdf.py:68-110.  ``idx`` a list (or 'all'): grads [B,in,len(idx)], one
        Jacobian column per listed (re-ordered) output; any other value ('closest', 'mindist'): grads [B,in,1] of the arg-min output.
        A single-output network takes the reference's scaled branch: grads [B,in] of the network-scale output w.r.t. the unscaled q."""
        if idx == "all":
            idx = list(range(self.out_channels))
        if self.out_channels == 1:
            qs = self._scale(q)
            y, g, _ = self._eng(qs.shape[0]).mlp_forward_vjp(self._rows(qs))
            grads = torch.from_numpy(g) / self.norm_dict["x"]["std"]
            return self._unscale(torch.from_numpy(y)), grads, torch.zeros(qs.shape[0])
        x = self._rows(q)
        order = np.asarray(self.order, dtype=np.int64)
        eng = self._eng(x.shape[0])
        if isinstance(idx, list):
            y, jac = eng.mlp_jacobian(x, order[np.asarray(idx, dtype=np.int64)])
            mi = np.argmin(y[:, order], axis=1)
            return torch.from_numpy(y[:, order]), torch.from_numpy(jac), torch.from_numpy(mi)
        if order.size == self.out_channels and (order == np.arange(self.out_channels)).all():
            y, g, mi = eng.mlp_forward_vjp(x)             # the arg-min of the raw outputs, found on the device
            return torch.from_numpy(y), torch.from_numpy(g).unsqueeze(2), torch.from_numpy(mi.astype(np.int64))
        # a link order of the caller's: the arg-min is over the re-ordered columns, the backward starts from that column
        y = self._rows(self.forward(x))
        dist = y[:, order]
        mi = np.argmin(dist, axis=1)
        need = np.unique(order[mi])
        grads = np.zeros((x.shape[0], x.shape[1], 1), np.float32)
        for c0 in range(0, need.size, 16):
            cols = need[c0:c0 + 16]
            _, jac = eng.mlp_jacobian(x, cols)
            for k, c in enumerate(cols):
                rows = order[mi] == c
                grads[rows, :, 0] = jac[rows, :, k]
        return torch.from_numpy(dist), torch.from_numpy(grads), torch.from_numpy(mi)

    def compute_signed_distance_wgrad2(self, q):
        """(dists, grads [B,in], minIdx) -- robot_sdf.py:144-151 (the vmap-of-vjp form of the same arg-min gradient)."""
        return self.functorch_vjp(q)

    def dist_grad_closest(self, q):
        """(dists [B,C], grads [B,in,1], minIdx [B]) -- robot_sdf.py:117-137; like the reference, at most the
        ``allocate_gradients`` row count is evaluated."""
        x = self._rows(q)
        if getattr(self, "maxInputSize", None):
            x = x[:min(self.maxInputSize, x.shape[0])]
        y, g, mi = self._eng(x.shape[0]).mlp_forward_vjp(x)
        return torch.from_numpy(y), torch.from_numpy(g).unsqueeze(2), torch.from_numpy(mi.astype(np.int64))

    def functorch_vjp(self, points):
        """(dists, grads [B,in], minIdx) -- robot_sdf.py:153-158."""
        x = self._rows(points)
        y, g, mi = self._eng(x.shape[0]).mlp_forward_vjp(x)
        return torch.from_numpy(y), torch.from_numpy(g), torch.from_numpy(mi.astype(np.int64))

    def dist_grad_closest_aot(self, q):
        return self.functorch_vjp(q)
