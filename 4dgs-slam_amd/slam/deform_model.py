"""Control-node deformation model of the dynamic branch -- the part of the reference's SC-GS model (gaussian_splatting/scene/
deform_model.py DeformModel, utils/time_utils.py ControlNodeWarp :788-1300) that the SLAM loops call: ``deform.step(x, time_input,
...)`` -> {d_xyz, d_rotation, d_scaling, d_opacity, d_color}, ``deform.deform.expand_time``, ``arap_loss`` / ``elastic_loss``, an
``optimizer``. The per-Gaussian half (K nearest nodes, RBF weights, blend and all chain rules) is the HIP kernel set behind
``control_nodes.node_blend`` (include/control_nodes.h); the per-NODE half is a small time-conditioned MLP (O(512) rows, torch).

Reduced on purpose: no hash-grid encoders, no node densification / pruning, no hyper-coordinates -- the shipped SLAM configuration
does not switch them on (arguments.py defaults) and the loops never call them."""
import math

import torch
from torch import nn

import control_nodes


def _embed(x, n_freq):
    """Positional encoding of utils/time_utils.py:208-273 (include_input, log-spaced sin / cos; order x, sin f0, cos f0, sin f1, ...):
    four launches whatever n_freq is."""
    if n_freq == 0:
        return x
    freqs = 2.0 ** torch.arange(n_freq, device=x.device, dtype=x.dtype)
    xf = x[..., None, :] * freqs[:, None]                                    # [..., F, C]
    sc = torch.stack([torch.sin(xf), torch.cos(xf)], -2)                      # [..., F, 2, C]
    return torch.cat([x, sc.flatten(-3)], -1)


def time_key(t):
    """Host-side name of a time sample: what ControlNodes.begin_iteration() files its batched evaluations under."""
    return round(float(t), 7)


def farthest_point_sample(xyz, npoint):
    """utils/time_utils.py:478-500 on one point set [N,3] -> indices [npoint] (deterministic start at index 0)."""
    N = xyz.shape[0]
    npoint = min(npoint, N)
    idx = torch.zeros(npoint, dtype=torch.long, device=xyz.device)
    dist = torch.full((N,), 1e10, device=xyz.device)
    far = torch.zeros((), dtype=torch.long, device=xyz.device)
    for i in range(npoint):
        idx[i] = far
        d = ((xyz - xyz[far]) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = torch.argmax(dist)
    return idx


class ControlNodes(nn.Module):
    def __init__(self, node_num=512, K=3, hidden=128, depth=4, t_multires=6, x_multires=6, d_rot_as_res=True, device="cuda"):
        super().__init__()
        self.K, self.max_nodes, self.d_rot_as_res = K, node_num, d_rot_as_res
        self.t_multires, self.x_multires = t_multires, x_multires
        self.device = torch.device(device)
        in_ch = 3 * (1 + 2 * x_multires) + (1 + 2 * t_multires)
        layers, c = [], in_ch
        for _ in range(depth):
            layers += [nn.Linear(c, hidden), nn.ReLU(inplace=True)]
            c = hidden
        self.trunk = nn.Sequential(*layers).to(self.device)
        self.head = nn.Linear(hidden, 10).to(self.device)           # d_xyz 3 | d_rotation 4 | d_scaling 3
        nn.init.zeros_(self.head.weight)                             # identity deformation at start (time_utils.py:401-404 normal_(1e-5))
        nn.init.zeros_(self.head.bias)
        self.nodes = nn.Parameter(torch.zeros(0, 3, device=self.device))
        self._node_radius = nn.Parameter(torch.zeros(0, device=self.device))
        self._node_weight = nn.Parameter(torch.zeros(0, 1, device=self.device))
        self.reg_loss = 0.0
        self.inited = False
        self._batch = None            # {time_key: [M,10] network output} of the current iteration (begin_iteration)
        self._graph = None

    node_num = property(lambda s: s.nodes.shape[0])

    @torch.no_grad()
    def init(self, init_pcl, **_):
        """ControlNodeWarp.init (:904-945): nodes by farthest-point sampling of the dynamic points, radius from the node spacing."""
        idx = farthest_point_sample(init_pcl.detach(), self.max_nodes)
        self.nodes = nn.Parameter(init_pcl.detach()[idx].clone())
        self._reset_radius()
        self.inited = True

    @torch.no_grad()
    def extend_node(self, init_pcl, **_):
        """:947-973: add nodes for new dynamic points up to the budget."""
        room = self.max_nodes - self.node_num
        if room <= 0 or init_pcl.shape[0] == 0:
            return
        idx = farthest_point_sample(init_pcl.detach(), room)
        self.nodes = nn.Parameter(torch.cat([self.nodes.detach(), init_pcl.detach()[idx]], 0))
        self._reset_radius()

    def _reset_radius(self):
        M = self.node_num
        if M > 1:
            kk = min(M - 1, 3)
            d = control_nodes.knn_points(self.nodes.detach()[None], self.nodes.detach()[None], K=kk + 1).dists[0, :, 1:]
            r = torch.sqrt(d.mean(-1).clamp_min(1e-8))
        else:
            r = torch.full((M,), 0.1, device=self.device)
        self._node_radius = nn.Parameter(torch.log(r))                # exp() activation, :893-895
        self._node_weight = nn.Parameter(torch.zeros(M, 1, device=self.device))     # sigmoid() activation, :897-898

    def expand_time(self, t):
        """:975-979."""
        return t.reshape(1, 1).expand(self.node_num, 1)

    # ---- one batched network evaluation per optimisation step ---------------------------------------------------------------
    # A dynamic mapping iteration (utils/slam_backend.py:336-771) asks the node network for 4-6 time samples per view (the view's
    # deltas, its flow partner's, the ARAP / elastic samples around it): ~60 evaluations of a 512-row MLP, ~40 launches each, plus
    # their backward. The nodes and weights only change at optimizer.step(), so all samples of an iteration are ONE batch.
    def begin_iteration(self, times):
        keys = sorted({time_key(t) for t in times})
        if not keys or self.node_num == 0:
            self._batch = None
            return
        M = self.node_num
        tt = torch.tensor(keys, dtype=torch.float32, device=self.device)[:, None]
        xe = _embed(self.nodes.detach(), self.x_multires)
        te = _embed(tt, self.t_multires)
        inp = torch.cat([xe[None].expand(len(keys), M, -1), te[:, None].expand(len(keys), M, -1)], -1)
        o = self.head(self.trunk(inp.reshape(len(keys) * M, -1))).reshape(len(keys), M, 10)
        self._batch = {k: o[i] for i, k in enumerate(keys)}
        self._graph = None

    def end_iteration(self):
        self._batch = None
        self._graph = None

    def node_deform(self, t, key=None):
        """:1038-1051: per-node translation / rotation / scale at time t [M,1] (key: the host-side value of t, see begin_iteration)."""
        o = self._batch.get(time_key(key)) if (key is not None and self._batch is not None) else None
        if o is None:
            h = self.trunk(torch.cat([_embed(self.nodes.detach(), self.x_multires), _embed(t, self.t_multires)], -1))
            o = self.head(h)
        return {"d_xyz": o[:, :3], "d_rotation": o[:, 3:7], "d_scaling": o[:, 7:10]}

    def forward(self, x, t, motion_mask=None, t_key=None, **_):
        """:1192-1258."""
        na = self.node_deform(t, t_key)
        out = control_nodes.node_blend(x, motion_mask, self.nodes, self._node_radius, self._node_weight, na["d_xyz"], na["d_rotation"],
                                       na["d_scaling"], None, K=min(self.K, self.node_num), d_rot_as_res=self.d_rot_as_res, raw=True)
        return {"d_xyz": out["d_xyz"], "d_rotation": out["d_rotation"], "d_scaling": out["d_scaling"], "d_opacity": None, "d_color": None}

    def _node_graph(self, K=4):
        if self._batch is not None and self._graph is not None:     # inside an iteration the nodes do not move
            return self._graph
        kk = min(self.node_num - 1, K)
        nb = control_nodes.knn_points(self.nodes.detach()[None], self.nodes.detach()[None], K=kk + 1).idx[0, :, 1:]
        if self._batch is not None:
            self._graph = nb
        return nb

    @staticmethod
    def sample_times(t, delta_t, t_samp_num=2):
        """The time samples arap_loss / elastic_loss evaluate around t (host floats): for begin_iteration."""
        arap = [t + (s / max(t_samp_num - 1, 1) - 0.5) * 2 * float(delta_t) for s in range(t_samp_num)]
        return arap + [t, t + float(delta_t)]

    def arap_loss(self, t=None, delta_t=0.05, t_samp_num=2, t_key=None, **_):
        """:1128-1141, reduced: edge lengths between neighbouring nodes are preserved between time t and t + delta_t."""
        if self.node_num < 2:
            return torch.zeros((), device=self.device)
        t0 = t.reshape(1, 1) if t is not None else torch.rand(1, 1, device=self.device)
        nb = self._node_graph()
        loss = 0.0
        ref = None
        for s in range(t_samp_num):
            off = (s / max(t_samp_num - 1, 1) - 0.5) * 2 * float(delta_t)
            ts = (t0 + off).expand(self.node_num, 1)
            p = self.nodes.detach() + self.node_deform(ts, None if t_key is None else t_key + off)["d_xyz"]
            e = (p[:, None] - p[nb]).norm(dim=-1)
            if ref is None:
                ref = e
            else:
                loss = loss + (e - ref).abs().mean()
        return loss

    def elastic_loss(self, t=None, delta_t=0.005, t_key=None, **_):
        """:1143-1165, reduced to a first-order smoothness of the node translations in time."""
        t0 = t.reshape(1, 1) if t is not None else torch.rand(1, 1, device=self.device)
        a = self.node_deform(t0.expand(self.node_num, 1), t_key)["d_xyz"]
        b = self.node_deform((t0 + float(delta_t)).expand(self.node_num, 1), None if t_key is None else t_key + float(delta_t))["d_xyz"]
        return (a - b).abs().mean()


class DeformModel:
    """gaussian_splatting/scene/deform_model.py:1-118: holder of the node warp + its optimizer."""

    def __init__(self, K=3, node_num=512, d_rot_as_res=True, lr=1e-3, device="cuda", **_):
        self.deform = ControlNodes(node_num=node_num, K=K, d_rot_as_res=d_rot_as_res, device=device)
        self.lr = lr
        self.optimizer = None
        self.reg_loss = 0.0

    def train_setting(self, *_):
        ps = [p for p in self.deform.parameters()]
        self.optimizer = torch.optim.Adam(ps, lr=self.lr, eps=1e-15)

    def step(self, x, time_input, iteration=0, feature=None, motion_mask=None, camera_center=None, time_interval=None, **kw):
        """deform_model.py:59-70."""
        return self.deform(x, time_input, motion_mask=motion_mask, t_key=kw.get("t_key"))

    def extend_node_from_point(self, init_pcl, **kw):
        if not self.deform.inited:
            self.deform.init(init_pcl)
        else:
            self.deform.extend_node(init_pcl)
        self.train_setting()
