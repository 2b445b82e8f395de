"""Rollout tensors that stay on the GPU until somebody reads them.

``MPPI.propagate()`` returns the reference's 5-tuple (MPPI.py:224) and sets ``all_traj / closest_dist_all / qdot / ...`` like the
reference does -- as ``LazyRollout`` objects.  What the reference's planner loop actually reads per iteration is a handful of rows
(``closests_dist_all[i, h]``, ``norm_basis[i, h]``, ``all_traj[best_idx:best_idx + 1]``, frankaPlanner.py:147-168); the candidate
search and the cost run on the device-resident copies.  So:

* indexing with a rollout index first (``x[i]``, ``x[i, h]``, ``x[i:j]`` for a short range, ``x[i, -1]``) fetches just those
  rollouts' rows (omds_get_rollout_rows: a few KB);
* anything else -- ``torch`` functions, arithmetic, comparisons, ``.numpy()``, ``.view()``, ``np.asarray`` ... -- materialises the
  whole tensor once (omds_get_rollouts) and behaves like the torch CPU tensor the reference would have returned.

Writing into one (``x[i] = v``, ``x += 1``) and pickling it (``pickle``, ``torch.save``, ZMQ ``send_pyobj``) materialise it first; a
written-to tensor no longer stands for the device copy (the candidate search then takes its host path).

A ``LazyRollout`` belongs to the propagate that made it: reading one after the next ``propagate()`` raises (the reference rebinds
fresh tensors every call, MPPI.py:86-91, so its callers never do that; ``.tensor()`` before the next call keeps a copy).  It is not a
``torch.Tensor`` subclass: ``MPPI.propagate()`` only hands these out when asked to (``lazy_rollouts=True``); the default is the
reference's plain tensors."""
from __future__ import annotations

import numpy as np
import torch

_ROW_FETCH_MAX = 64      # a leading slice / index list of up to this many rollouts is served by a row fetch


class LazyRollout:
    def __init__(self, owner, key, shape, generation):
        self._owner, self._key, self._shape, self._gen = owner, key, tuple(shape), generation
        self._full = None
        self._dirty = False       # written into by the caller: no longer the device copy

    # ---- materialisation ---------------------------------------------------------------------------
    def _check(self):
        if self._full is None and self._gen != self._owner._generation:
            raise RuntimeError(f"{self._key} of an earlier propagate() was never read and its device copy has been overwritten: "
                               f"call .tensor() before the next propagate() to keep it")

    def tensor(self) -> torch.Tensor:
        """The whole tensor as a torch CPU tensor (one fetch of all rollout tensors, cached by the owner)."""
        if self._full is None:
            self._check()
            self._full = self._owner._fetch()[self._key]
        return self._full

    def _rows(self, t):
        self._check()
        return torch.from_numpy(self._owner._engine.get_rollout_rows(t, want=(self._key,))[self._key])

    # ---- tensor look-alike -------------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size(self._shape)

    def size(self, dim=None):
        return self.shape if dim is None else self._shape[dim]

    def dim(self):
        return len(self._shape)

    ndim = property(lambda self: len(self._shape))
    dtype = torch.float32
    device = torch.device("cpu")

    def __len__(self):
        return self._shape[0]

    def __getitem__(self, idx):
        if self._full is not None:
            return self._full[idx]
        lead, rest = (idx[0], idx[1:]) if isinstance(idx, tuple) else (idx, ())
        N = self._shape[0]
        rows = None
        if isinstance(lead, (int, np.integer)) or (isinstance(lead, torch.Tensor) and lead.ndim == 0 and not lead.dtype.is_floating_point
                                                   and lead.dtype != torch.bool):
            i = int(lead)
            rows, squeeze = [i + N if i < 0 else i], True
        elif isinstance(lead, slice):
            r = range(*lead.indices(N))
            if 0 < len(r) <= _ROW_FETCH_MAX:
                rows, squeeze = list(r), False
        if rows is None or not all(0 <= i < N for i in rows):
            return self.tensor()[idx]
        out = self._rows(rows)
        out = out[0] if squeeze else out
        return out[rest if squeeze else (slice(None),) + tuple(rest)] if rest else out

    def __setitem__(self, idx, value):
        self._dirty = True
        self.tensor()[idx] = value.tensor() if isinstance(value, LazyRollout) else value

    def __reduce__(self):                 # pickle / torch.save / send_pyobj: the plain tensor travels
        return (_identity, (self.tensor(),))

    def __array__(self, dtype=None, copy=None):
        a = self.tensor().numpy()
        return a.astype(dtype) if dtype is not None else a

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        conv = lambda a: a.tensor() if isinstance(a, LazyRollout) else a
        args = tuple([conv(x) for x in a] if isinstance(a, (list, tuple)) else conv(a) for a in args)
        kwargs = {k: conv(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)

    def __getattr__(self, name):          # .numpy(), .view(), .flatten(), .min(), ... : the materialised tensor's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __repr__(self):
        return f"LazyRollout({self._key}, shape={self._shape}, {'fetched' if self._full is not None else 'on device'})"

    def __iter__(self):
        return iter(self.tensor())

    def __bool__(self):
        return bool(self.tensor())


def _identity(t):
    return t


def _inplace(name):
    def f(self, other):
        self._dirty = True
        getattr(self.tensor(), name)(other.tensor() if isinstance(other, LazyRollout) else other)
        return self
    f.__name__ = name
    return f


def _delegate(name):
    def f(self, *a):
        return getattr(self.tensor(), name)(*[x.tensor() if isinstance(x, LazyRollout) else x for x in a])
    f.__name__ = name
    return f


for _n in ("add", "sub", "mul", "truediv", "floordiv", "pow", "matmul", "mod", "and", "or", "xor"):
    setattr(LazyRollout, f"__{_n}__", _delegate(f"__{_n}__"))
    setattr(LazyRollout, f"__r{_n}__", _delegate(f"__r{_n}__"))
for _n in ("iadd", "isub", "imul", "itruediv"):
    setattr(LazyRollout, f"__{_n}__", _inplace(f"__{_n}__"))
for _n in ("lt", "le", "gt", "ge", "eq", "ne", "neg", "abs", "invert"):
    setattr(LazyRollout, f"__{_n}__", _delegate(f"__{_n}__"))
LazyRollout.__hash__ = object.__hash__      # __eq__ is elementwise like a tensor's; identity hashing like a tensor's
