"""SDF training on the GPU -- the per-epoch work of ``mlp_learn/train_sdf.py:96-151`` (full-batch forward, MSE, backward, Adam)
behind ``omds_trainer_*`` (csrc/train.hip), plus the host-side pieces of that script that are O(1) per epoch: the
``ReduceLROnPlateau`` schedule and the checkpoint dictionary in the reference's format."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class SdfTrainer:
    """``dims`` = [3 * (raw inputs), hidden ..., out_channels]; weights ``W[i]`` are [out, in] like torch ``nn.Linear``."""

    def __init__(self, dims, act="relu", device=0, lib=None):
        self.lib = lib if lib is not None else L.load()
        self.dims = [int(v) for v in dims]
        self.nl = len(self.dims) - 1
        self.act = act
        d = np.asarray(self.dims, dtype=np.int32)
        h = C.c_void_p()
        rc = self.lib.omds_trainer_create(int(device), self.nl, L.iptr(d), 0 if act == "relu" else 1, C.byref(h))
        if rc != 0:
            raise L.OmdsError(f"omds_trainer_create failed ({rc}): {(self.lib.omds_trainer_last_error(None) or b'?').decode()}")
        self.h = h
        self.B = 0
        self.Bv = 0
        self.hyper = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8)     # what the last step() used (the checkpoint's param_groups)

    def _ck(self, rc):
        if rc != 0:
            raise L.OmdsError(f"omds trainer error {rc}: {(self.lib.omds_trainer_last_error(self.h) or b'?').decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.omds_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, W, b):
        Ws = [L.f32(w).reshape(self.dims[i + 1], self.dims[i]) for i, w in enumerate(W)]
        bs = [L.f32(v).reshape(self.dims[i + 1]) for i, v in enumerate(b)]
        Wp = (L.F32P * self.nl)(*[L.fptr(w) for w in Ws])
        bp = (L.F32P * self.nl)(*[L.fptr(v) for v in bs])
        self._ck(self.lib.omds_trainer_set_weights(self.h, Wp, bp))

    def get_weights(self):
        Ws = [np.zeros((self.dims[i + 1], self.dims[i]), np.float32) for i in range(self.nl)]
        bs = [np.zeros(self.dims[i + 1], np.float32) for i in range(self.nl)]
        Wp = (L.F32P * self.nl)(*[L.fptr(w) for w in Ws])
        bp = (L.F32P * self.nl)(*[L.fptr(v) for v in bs])
        self._ck(self.lib.omds_trainer_get_weights(self.h, Wp, bp))
        return Ws, bs

    def set_data(self, x, y):
        x = L.f32(x).reshape(-1, self.dims[0] // 3)
        y = L.f32(y).reshape(x.shape[0], self.dims[-1])
        self._ck(self.lib.omds_trainer_set_data(self.h, L.fptr(x), L.fptr(y), x.shape[0]))
        self.B = x.shape[0]

    def set_val_data(self, x, y):
        """The validation split (train_sdf.py:84-86): lives beside the training set, evaluated by ``eval(val=True)`` with the
        trainer's current weights."""
        x = L.f32(x).reshape(-1, self.dims[0] // 3)
        y = L.f32(y).reshape(x.shape[0], self.dims[-1])
        self._ck(self.lib.omds_trainer_set_val_data(self.h, L.fptr(x), L.fptr(y), x.shape[0]))
        self.Bv = x.shape[0]

    def step(self, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        """One epoch of train_sdf.py:105-113; returns the loss before the update."""
        loss = C.c_float()
        self._ck(self.lib.omds_trainer_step(self.h, float(lr), float(betas[0]), float(betas[1]), float(eps), C.cast(C.byref(loss), L.F32P)))
        self.hyper = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps))
        return loss.value

    def eval(self, want_pred=False, val=False):
        """Forward + mse_loss with the current weights, no update, on the training set or (``val``) the validation set."""
        mse = C.c_float()
        pred = np.zeros((self.Bv if val else self.B, self.dims[-1]), np.float32) if want_pred else None
        self._ck(self.lib.omds_trainer_eval(self.h, 1 if val else 0, C.cast(C.byref(mse), L.F32P), L.fptr(pred)))
        return (mse.value, pred) if want_pred else mse.value

    # ---- torch.optim.Adam's state, in torch's own state_dict() shape (train_sdf.py:130-138 saves it; a resumed run loads it) ----
    def _state_arrays(self):
        mk = lambda: ([np.zeros((self.dims[i + 1], self.dims[i]), np.float32) for i in range(self.nl)],
                      [np.zeros(self.dims[i + 1], np.float32) for i in range(self.nl)])
        (mW, mb), (vW, vb) = mk(), mk()
        return mW, mb, vW, vb

    def optimizer_state_dict(self):
        """{'state': {p: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} with the parameters in ``model.parameters()``
        order (weight 0, bias 0, weight 1, ...) -- what ``torch.optim.Adam.state_dict()`` returns and ``load_state_dict`` takes."""
        import torch
        mW, mb, vW, vb = self._state_arrays()
        ptr = lambda arrs: (L.F32P * self.nl)(*[L.fptr(a) for a in arrs])
        step = C.c_int64()
        self._ck(self.lib.omds_trainer_get_optimizer_state(self.h, ptr(mW), ptr(mb), ptr(vW), ptr(vb), C.byref(step)))
        state = {}
        if step.value > 0:          # torch creates a parameter's state at its first step
            for i in range(self.nl):
                for j, (m, v) in enumerate(((mW[i], vW[i]), (mb[i], vb[i]))):
                    state[2 * i + j] = {"step": torch.tensor(float(step.value)), "exp_avg": torch.from_numpy(m), "exp_avg_sq": torch.from_numpy(v)}
        group = {"lr": self.hyper["lr"], "betas": tuple(self.hyper["betas"]), "eps": self.hyper["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(2 * self.nl))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Restores exp_avg / exp_avg_sq / step from a torch-Adam state dict (this class's or the reference's own checkpoint)."""
        mW, mb, vW, vb = self._state_arrays()
        st = sd.get("state", {}) if sd else {}
        step = 0
        for i in range(self.nl):
            for j, (m, v) in enumerate(((mW[i], vW[i]), (mb[i], vb[i]))):
                e = st.get(2 * i + j)
                if e is None:
                    continue
                m[...] = np.asarray(e["exp_avg"], dtype=np.float32).reshape(m.shape)
                v[...] = np.asarray(e["exp_avg_sq"], dtype=np.float32).reshape(v.shape)
                step = max(step, int(float(e["step"])))
        ptr = lambda arrs: (L.F32P * self.nl)(*[L.fptr(a) for a in arrs])
        self._ck(self.lib.omds_trainer_set_optimizer_state(self.h, ptr(mW), ptr(mb), ptr(vW), ptr(vb), step))
        if sd and sd.get("param_groups"):
            g = sd["param_groups"][0]
            self.hyper = dict(lr=float(g["lr"]), betas=tuple(float(x) for x in g["betas"]), eps=float(g["eps"]))


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold_mode='rel') as train_sdf.py:85-87 configures it."""

    def __init__(self, lr, factor=0.5, patience=5000, threshold=0.01, cooldown=0, min_lr=0.0, eps=1e-4):
        self.lr, self.factor, self.patience, self.threshold = float(lr), factor, patience, threshold
        self.cooldown, self.min_lr, self.eps = cooldown, min_lr, eps
        self.best, self.num_bad, self.cooldown_counter = float("inf"), 0, 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad = 0
        if self.num_bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.cooldown_counter, self.num_bad = self.cooldown, 0
        return self.lr


def checkpoint_dict(epoch, W, b, n_in_raw, n_out, optimizer_state=None):
    """The dictionary train_sdf.py:130-138 saves (keys of a skip-less MLPRegression: layers.0.<i>.0.weight / bias; identity
    normalisation because of the NeRF features, train_sdf.py:78-82), loadable by RobotSdfCollisionNet.load_weights.
    ``optimizer_state`` = ``SdfTrainer.optimizer_state_dict()`` (the reference saves ``optimizer.state_dict()`` there)."""
    import torch
    sd = {}
    for i, (w, v) in enumerate(zip(W, b)):
        sd[f"layers.0.{i}.0.weight"] = torch.from_numpy(np.ascontiguousarray(w))
        sd[f"layers.0.{i}.0.bias"] = torch.from_numpy(np.ascontiguousarray(v))
    return {"epoch": int(epoch), "model_state_dict": sd, "optimizer_state_dict": optimizer_state,
            "norm": {"x": {"mean": torch.zeros(n_in_raw), "std": torch.ones(n_in_raw)},
                     "y": {"mean": torch.zeros(n_out), "std": torch.ones(n_out)}}}


def planar_link_distances(rng, batch, n_links=2, link_len=3.0, reach=None):
    """A synthetic SDF data set in the layout of the reference's 2-D toy data (train_sdf.py:38-44: x = [q, point], y = link
    distances): a planar chain of ``n_links`` links of length ``link_len`` (standalonePlanar2d.py:67-69), joint angles uniform
    in [-pi, pi], points uniform in the disc the arm can reach (+ 20 %), y[:, c] = distance from the point to link c's segment."""
    reach = reach or 1.2 * n_links * link_len
    q = rng.uniform(-np.pi, np.pi, (batch, n_links))
    p = rng.uniform(-reach, reach, (batch, 2))
    ang = np.cumsum(q, axis=1)
    joints = np.concatenate((np.zeros((batch, 1, 2)), np.cumsum(link_len * np.stack((np.cos(ang), np.sin(ang)), -1), axis=1)), axis=1)
    y = np.zeros((batch, n_links))
    for c in range(n_links):
        a, b = joints[:, c], joints[:, c + 1]
        ab, ap = b - a, p - a
        t = np.clip((ap * ab).sum(-1) / (ab * ab).sum(-1), 0.0, 1.0)
        y[:, c] = np.linalg.norm(ap - t[:, None] * ab, axis=1)
    return np.concatenate((q, p), axis=1).astype(np.float32), y.astype(np.float32)
