"""SEDS nominal DS -- mirrors ``ds_mppi/functions/SEDS.py`` (class SEDS, lines 8-74): a Gaussian mixture regression from
x = q - q_goal to the nominal velocity, read from the reference's ``content/ds/*.mat`` files (Mu, Sigma, Priors, xT).

Inside ``MPPI.propagate`` the velocity is evaluated on the GPU (``modulate_core``: ``omds_set_ds_seds``); this class is the
parameter holder a driver constructs (``DS1 = SEDS('content/ds/seds_sine10.mat', q_0.unsqueeze(1))``,
frankaIntegrator.py:70-71) and derives the per-component quantities the way the reference does (``torch.inverse`` /
``torch.det`` in float32), so that the device sees the reference's numbers."""
import numpy as np
import torch


class SEDS:
    def __init__(self, fname, attr=None):
        if isinstance(fname, dict):                      # already-loaded arrays (tests: fixture data)
            data = fname
        else:
            from scipy.io import loadmat
            data = loadmat(fname)
        self.dtype = torch.float32
        self.Mu = torch.tensor(np.asarray(data["Mu"])).to(self.dtype)
        self.Sigma = torch.tensor(np.asarray(data["Sigma"])).to(self.dtype)
        self.Priors = torch.tensor(np.asarray(data["Priors"])).to(self.dtype)
        self.q_goal = torch.tensor(np.asarray(data["xT"])).to(self.dtype)
        if attr is not None:
            self.q_goal = torch.as_tensor(attr).to(self.dtype)
        self.dof = int(self.Mu.shape[0] / 2)
        self.n_gaussians = self.Sigma.shape[2]
        d = self.dof
        self.Sigma_inv = torch.zeros([d, d, self.n_gaussians]).to(self.dtype)
        self.det = torch.zeros([self.n_gaussians]).to(self.dtype)
        for i in range(self.n_gaussians):                # SEDS.py:22-24
            self.Sigma_inv[:, :, i] = torch.inverse(self.Sigma[:d, :d, i]).to(self.dtype)
            self.det[i] = torch.abs(torch.det(self.Sigma[:d, :d, i])).to(self.dtype)
        self.seds_thr = 1e-2
        self.lin_thr = 1e-2

    def device_params(self):
        """(mu_in [G,n], b [G,n], sigma_inv [G,n,n], A [G,n,n], prior [G], den [G]) for ``Engine.set_ds_seds``:
        gaussPDF inverts Sigma again per call (SEDS.py:31) and GMR multiplies Sigma[out,in] @ Sigma_inv first (SEDS.py:53-55)."""
        d, G = self.dof, self.n_gaussians
        mu_in = self.Mu[:d, :].t().contiguous()
        b = self.Mu[d:, :].t().contiguous()
        s_inv = torch.stack([torch.inverse(self.Sigma[:d, :d, j]) for j in range(G)])
        A = torch.stack([self.Sigma[d:, :d, j] @ self.Sigma_inv[:d, :d, j] for j in range(G)])
        den = torch.stack([torch.sqrt((2 * torch.tensor(torch.pi) ** d) * self.det[j] + torch.tensor(1e-100)) for j in range(G)])
        return tuple(t.numpy().astype(np.float32) for t in (mu_in, b, s_inv, A, self.Priors.reshape(-1), den))

    def get_velocity(self, x):
        """Host convenience with the reference's semantics (SEDS.py:59-74); the rollouts use the device kernel."""
        x = torch.as_tensor(x, dtype=torch.float32)
        d, G = self.dof, self.n_gaussians
        mu_in, b, s_inv, A, prior, den = (torch.from_numpy(t) for t in self.device_params())
        xd = x - self.q_goal.reshape(1, d)
        dd = xd[:, None, :] - mu_in[None]                                        # [N, G, d]
        prob = torch.einsum("ngr,grc,ngc->ng", dd, s_inv, dd)
        pxi = prior[None] * (torch.exp(-0.5 * prob) / den[None])      # the reference's order: it matters where exp() is denormal
        beta = torch.clamp((pxi / pxi.sum(dim=1, keepdim=True)).nan_to_num(), min=1e-8)
        y = (beta[:, :, None] * (b[None] + torch.einsum("grc,ngc->ngr", A, dd))).sum(dim=1)
        dst, yn = xd.norm(dim=1), y.norm(dim=1)
        far, weak = dst > self.lin_thr, yn < self.seds_thr
        out = y.clone()
        out[far] = y[far] / yn[far, None]
        lin = far & weak
        out[lin] = -xd[lin] / dst[lin, None]
        return out
