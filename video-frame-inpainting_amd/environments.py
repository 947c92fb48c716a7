"""Evaluation and training environments of the bi-TAI path (reference src/environments/environments.py:55-485).

Same object protocol as the reference so ``predict.py`` / ``train.py``-style drivers read the same:
  eval:   env = create_eval_environment(model, checkpoints_dir, name, snapshot_file_name, padding_size)
          env.set_test_inputs(P, F); env.T = T; env.eval(); env.forward_test(); env.gen_output['pred']
  train:  env = create_training_environment(...); env.set_train_inputs(P, F, GT); env.K, env.T, env.F = ...
          env.train(); env.forward_train(); env.optimize_parameters(); env.get_current_errors()
Checkpoints keep the reference's dict layout {updates, sum_avg_psnr_err, sum_avg_ssim_err, generator, optimizer_G,
discriminator, optimizer_D} (environments.py:178-194, 290-297); evaluation loads ``snapshot['generator']`` only (:113).

What differs (same arithmetic):
  * no ``Variable`` / ``volatile``: inference runs under ``torch.no_grad()`` and, with ``use_graph=True``, as a replayed
    hipGraph (graph.py) keyed by the input shapes;
  * the device is explicit, so one process per GPU can own ``cuda:LOCAL_RANK``;
  * ``optimize_parameters`` all-reduces generator and discriminator gradients across data-parallel ranks (parallel.py)
    right after each backward, in the reference's G-then-D order; with one process it is the reference's step.
"""
import os

import numpy as np
import torch

from . import conv_ops, parallel
from .graph import GraphedForward
from .losses import GDL
from .mcnet import MCNetFillInModel
from .sn_discriminator import SNDiscriminator
from .ablations import (BidirectionalSimpleAverageFillInModel, BidirectionalTimeWeightedAverageFillInModel,
                        TimeWeightedInterpolationFillInModel, TimeWeightedPFFillInModel)
from .tai import TAIFillInModel
from .util import inverse_transform, move_to_devices, weights_init


def create_eval_environment(fill_in_model, checkpoints_dir, name, snapshot_file_name, padding_size, device=None,
                            load_snapshot=True, use_graph=False):
    env = BaseVideoFillInEnvironment(fill_in_model, checkpoints_dir, name, padding_size, device=device,
                                     use_graph=use_graph)
    if load_snapshot and not isinstance(fill_in_model, TimeWeightedPFFillInModel):   # environments.py:57-58
        env.load(snapshot_file_name)
    print('Loaded evaluation environment')
    return env


def create_training_environment(fill_in_model, c_dim, checkpoints_dir, name, max_K, max_T, max_F, image_size, alpha,
                                beta, lr, beta1, df_dim, Ip, disc_window_size, padding_size, device=None, graph_step=False):
    if isinstance(fill_in_model, (TAIFillInModel, TimeWeightedInterpolationFillInModel,
                                  BidirectionalSimpleAverageFillInModel, BidirectionalTimeWeightedAverageFillInModel)):
        env = TAITrainingEnvironment(      # environments.py:29-31
            fill_in_model, checkpoints_dir, name, image_size, c_dim, alpha, beta, lr, beta1,
                                     df_dim, Ip, disc_window_size, max_K, max_T, max_F, padding_size, device=device,
                                     graph_step=graph_step)
    elif isinstance(fill_in_model, MCNetFillInModel):
        env = MCNetTrainingEnvironment(fill_in_model, checkpoints_dir, name, image_size, c_dim, alpha, beta, lr, beta1,
                                       df_dim, Ip, disc_window_size, max_K, max_T, max_F, padding_size, device=device,
                                       graph_step=graph_step)
    else:
        raise RuntimeError('Tried to create a training environment for object of unsupported type %s'
                           % type(fill_in_model).__name__)
    if os.path.isfile(os.path.join(checkpoints_dir, name, 'model_latest.ckpt')):
        print('Loading latest snapshot...')
        env.load('model_latest.ckpt')
    print('Loaded training environment')
    return env


class _parameters_frozen(object):
    """``with _parameters_frozen(module):`` -- the module's parameters do not require grad inside the block."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)
        return False


class BaseVideoFillInEnvironment(object):
    """environments.py:64-119."""

    def __init__(self, video_fill_in_model, checkpoints_dir, name, padding_size, device=None, use_graph=False):
        self.save_dir = os.path.join(checkpoints_dir, name)
        self.padding_size = padding_size
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.generator = move_to_devices(video_fill_in_model, self.device)
        self.generator.apply(weights_init)
        self.K = self.T = self.F = None
        self.use_graph = use_graph
        self._graphs = {}

    def set_test_inputs(self, preceding_frames, following_frames):
        self.preceding_frames = preceding_frames.contiguous().to(self.device, non_blocking=True)
        self.following_frames = following_frames.contiguous().to(self.device, non_blocking=True)

    def set_gt_middle_frames_test(self, gt_middle_frames):
        self.gt_middle_frames = gt_middle_frames.contiguous().to(self.device, non_blocking=True)

    def forward_test(self):
        if self.use_graph:
            key = (self.T, tuple(self.preceding_frames.shape), tuple(self.following_frames.shape))
            if key not in self._graphs:
                self._graphs[key] = GraphedForward(self.generator, self.T, self.preceding_frames, self.following_frames)
            self.gen_output = self._graphs[key](self.preceding_frames, self.following_frames)
        else:
            with torch.no_grad():
                self.gen_output = self.generator(self.T, self.preceding_frames, self.following_frames)

    def load(self, snapshot_file_name):
        save_path = os.path.join(self.save_dir, snapshot_file_name)
        if os.path.isfile(save_path):
            print('=> loading snapshot from {}'.format(save_path))
            try:
                snapshot = torch.load(save_path, map_location=self.device, weights_only=False)
            except UnicodeDecodeError:
                # the published checkpoints were pickled by Python 2.7 / torch 0.3.1 (bashes/download/
                # download_model_checkpoints.bash): their byte strings need latin1
                snapshot = torch.load(save_path, map_location=self.device, weights_only=False, encoding='latin1')
        else:
            raise RuntimeError('Failed to find snapshot at path %s' % save_path)
        self.generator.load_state_dict(snapshot['generator'])
        conv_ops.invalidate_derived(self.generator)
        self._graphs.clear()
        return snapshot

    def eval(self):
        self.generator.eval()


class BaseTrainingEnvironment(BaseVideoFillInEnvironment):
    """environments.py:122-259."""

    STEP_GRAPH_WARMUP = 2      # eager updates per (K, T, F, batch shape) before the update is captured
    MAX_STEP_GRAPHS = 2        # a captured update keeps its activations (tens of GB at cfg3): further shapes (--sample_KTF) stay eager

    def __init__(self, fill_in_model, checkpoints_dir, name, lr, beta1, max_K, max_T, max_F, padding_size, device=None,
                 graph_step=False):
        super().__init__(fill_in_model, checkpoints_dir, name, padding_size, device=device)
        # graph_step: ``train_step`` captures one whole update (forward, both backward passes, both Adam steps: ~10,000
        # kernel launches) as a hipGraph after STEP_GRAPH_WARMUP eager updates and replays it; Adam then keeps its step
        # counter on the device (``capturable``).  Off by default: the eager sequence is the reference's.
        self.graph_step = bool(graph_step)
        self._step_graphs = {}
        self.start_update = 0
        self.total_updates = 0
        self.start_sum_avg_psnr_err = 0
        self.start_sum_avg_ssim_err = 0
        self.max_K, self.max_T, self.max_F = max_K, max_T, max_F
        self.optimizer_G = torch.optim.Adam(self.generator.parameters(), lr=lr, betas=(beta1, 0.999), capturable=self.graph_step)
        self._reducer_G = parallel.GradAllReducer(self.generator.parameters())
        # The reference draws (K, T, F) from the global numpy RNG (environments.py:417-427).  Data-parallel replicas must
        # draw the SAME values every step, and anything else that touches the global RNG on one rank only (a data-loader
        # retry, user code) would make them diverge for the rest of the run: a private stream, seeded identically on
        # every rank (``seed_KTF`` to change it).
        self._ktf_rng = np.random.RandomState(0)

    def seed_KTF(self, seed):
        self._ktf_rng = np.random.RandomState(seed)

    def sample_KTF(self, allow_random_sampling):
        if allow_random_sampling:
            K = self._ktf_rng.randint(1, self.max_K + 1)
            T = self._ktf_rng.randint(1, self.max_T + 1)
            F = self._ktf_rng.randint(1, self.max_F + 1)
        else:
            K, T, F = self.max_K, self.max_T, self.max_F
        return K, T, F

    def set_train_inputs(self, preceding_frames, following_frames, gt_middle_frames):
        self.preceding_frames = preceding_frames.contiguous().to(self.device, non_blocking=True)
        self.following_frames = following_frames.contiguous().to(self.device, non_blocking=True)
        self.gt_middle_frames = gt_middle_frames.contiguous().to(self.device, non_blocking=True)

    def forward_train(self):
        self.gen_output = self.generator(self.T, self.preceding_frames, self.following_frames)

    # attributes an update produces (tensors of the captured graph's pool when the update is replayed)
    _STEP_OUTPUTS = ('gen_output', 'loss_G', 'Lp', 'gdl', 'L_GAN', 'loss_d_fake', 'loss_d_real', 'loss_D', 'Lp_forward',
                     'Lp_backward', 'gdl_forward', 'gdl_backward')

    def train_step(self, preceding_frames, following_frames, gt_middle_frames):
        """One update on a batch: set_train_inputs + forward_train + optimize_parameters (src/train.py:147-169), with
        ``self.K, self.T, self.F`` set by the caller.  With ``graph_step`` (one process; data-parallel runs stay eager) the
        update is captured once per (K, T, F, batch shape) and replayed from static input buffers."""
        if not self.graph_step or parallel.world_size() > 1:
            self.set_train_inputs(preceding_frames, following_frames, gt_middle_frames)
            self.forward_train()
            self.optimize_parameters()
            return
        key = (self.K, self.T, self.F, tuple(preceding_frames.shape), tuple(following_frames.shape), tuple(gt_middle_frames.shape))
        state = self._step_graphs.setdefault(key, {'eager': 0})
        captured = sum(1 for v in self._step_graphs.values() if 'graph' in v)
        if state['eager'] < self.STEP_GRAPH_WARMUP or ('graph' not in state and captured >= self.MAX_STEP_GRAPHS):
            # MIOpen's algorithm search, lazy allocations, Adam's state -- or no room for another captured update
            state['eager'] += 1
            self.set_train_inputs(preceding_frames, following_frames, gt_middle_frames)
            self.forward_train()
            self.optimize_parameters()
            return
        if 'graph' not in state:
            self.set_train_inputs(preceding_frames, following_frames, gt_middle_frames)
            state['inputs'] = (self.preceding_frames.clone(), self.following_frames.clone(), self.gt_middle_frames.clone())
            self.preceding_frames, self.following_frames, self.gt_middle_frames = state['inputs']
            self._prepare_capture()
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            # No tensor that carries autograd history may be released inside the capture or outlive it: the previous
            # (eager) update's outputs are detached before, this update's before the capture ends.  Either one left alone
            # ends in a fault inside the HIP runtime when the capture is closed (tools/graph_op_bisect.py gen_keep_out*,
            # tools/graph_step_bisect.py m0).
            self._detach_step_outputs()
            # ... and whatever autograd graph pieces are only kept alive by reference cycles go NOW, not at some allocation
            # inside the capture.  What this cannot see is a caller that still holds a tensor with history from an earlier eager
            # update (a loss, ``env.gen_output['pred']``): the capturing call is made with such references dropped or detached.
            import gc
            gc.collect()
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('train_step: already inside a stream capture; a step graph cannot be nested')
            with torch.cuda.graph(graph):
                self.forward_train()
                self.optimize_parameters()
                self._detach_step_outputs()
            state['graph'] = graph
            state['outputs'] = {k: getattr(self, k) for k in self._STEP_OUTPUTS if hasattr(self, k)}
        else:
            p, f, g = state['inputs']
            p.copy_(preceding_frames, non_blocking=True)
            f.copy_(following_frames, non_blocking=True)
            g.copy_(gt_middle_frames, non_blocking=True)
            self.preceding_frames, self.following_frames, self.gt_middle_frames = p, f, g
            for k, v in state['outputs'].items():
                setattr(self, k, v)
        state['graph'].replay()
        # the replay recomputed the derived weights (Winograd-domain filters) BEFORE its optimizer steps and moved no
        # version counter: whatever runs eagerly next (validation, a snapshot's forward) must rebuild them
        conv_ops.invalidate_derived()

    def _detach_step_outputs(self):
        for k in self._STEP_OUTPUTS:
            v = getattr(self, k, None)
            if isinstance(v, dict):
                setattr(self, k, {name: t.detach() for name, t in v.items()})
            elif torch.is_tensor(v):
                setattr(self, k, v.detach())

    def _prepare_capture(self):
        """Host-side work a captured update may not do (host-to-device copies): done here, once."""

    def get_current_state_dict(self, total_updates, sum_avg_psnr_err, sum_avg_ssim_err):
        return {
            'updates': total_updates,
            'sum_avg_psnr_err': sum_avg_psnr_err,
            'sum_avg_ssim_err': sum_avg_ssim_err,
            'generator': self.generator.state_dict(),
            'optimizer_G': self.optimizer_G.state_dict(),
        }

    def load(self, snapshot_file_name):
        snapshot = super().load(snapshot_file_name)
        self.start_update = snapshot['updates']
        self.start_sum_avg_psnr_err = snapshot['sum_avg_psnr_err']
        self.start_sum_avg_ssim_err = snapshot['sum_avg_ssim_err']
        self.optimizer_G.load_state_dict(snapshot['optimizer_G'])
        self._step_graphs.clear()
        return snapshot

    def save(self, snapshot_file_name, total_updates, sum_avg_psnr_err, sum_avg_ssim_err):
        if parallel.rank() != 0:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(self.get_current_state_dict(total_updates, sum_avg_psnr_err, sum_avg_ssim_err),
                   os.path.join(self.save_dir, snapshot_file_name))

    def _zero_grad(self, optimizer, reducer):
        """One process: the reference's ``optimizer.zero_grad()``.  Data parallel: gradients live in the reducer's flat
        buckets and each bucket's all-reduce starts during the backward pass (parallel.GradAllReducer)."""
        if parallel.world_size() > 1:
            reducer.zero_grad()
        else:
            optimizer.zero_grad()

    def optimize_parameters(self):
        self._zero_grad(self.optimizer_G, self._reducer_G)
        self.compute_loss_G()
        self.loss_G.backward()
        self._reducer_G.allreduce_()
        self.optimizer_G.step()

    def compute_loss_G(self):
        self.loss_G = torch.zeros(1, device=self.device)

    def get_current_errors(self):
        return {'G_loss': float(self.loss_G.item())}

    def train(self):
        self.generator.train()


class L2GDLDiscTrainingEnvironment(BaseTrainingEnvironment):
    """environments.py:262-397: loss_G = alpha (MSE + GDL)(pred) + beta BCE(D(cat[P, pred, F]), 1);
    loss_D = BCE(D(fake.detach()), window labels) + BCE(D(real), 1)."""

    def __init__(self, fill_in_model, checkpoints_dir, name, image_size, c_dim, alpha, beta, lr, beta1, df_dim, Ip,
                 disc_t, max_K, max_T, max_F, padding_size, device=None, graph_step=False):
        super().__init__(fill_in_model, checkpoints_dir, name, lr, beta1, max_K, max_T, max_F, padding_size, device=device,
                         graph_step=graph_step)
        self._fake_labels = {}
        self.loss_Lp = torch.nn.MSELoss()
        self.loss_gdl = GDL()
        self.loss_d = torch.nn.BCEWithLogitsLoss()
        self.alpha, self.beta, self.disc_t = alpha, beta, disc_t
        discriminator = SNDiscriminator((image_size[0] + padding_size[0], image_size[1] + padding_size[1]), c_dim,
                                        disc_t, df_dim, Ip)
        discriminator = move_to_devices(discriminator, self.device)
        discriminator.apply(weights_init)
        self.discriminator = discriminator
        self.optimizer_D = torch.optim.Adam(self.discriminator.parameters(), lr=lr, betas=(beta1, 0.999), capturable=self.graph_step)
        self._reducer_D = parallel.GradAllReducer(self.discriminator.parameters())

    def sync_replicas(self):
        """Data-parallel start-up: identical weights and identical SN ``u`` vectors on every rank (rank 0's)."""
        parallel.materialise_sn_vectors(self.discriminator)
        parallel.broadcast_module_state(self.generator)
        parallel.broadcast_module_state(self.discriminator)

    def get_current_state_dict(self, total_updates, sum_avg_psnr_err, sum_avg_ssim_err):
        state = super().get_current_state_dict(total_updates, sum_avg_psnr_err, sum_avg_ssim_err)
        state['discriminator'] = self.discriminator.state_dict()
        state['optimizer_D'] = self.optimizer_D.state_dict()
        return state

    def load(self, snapshot_file_name):
        snapshot = super().load(snapshot_file_name)
        self.discriminator.load_state_dict(snapshot['discriminator'])
        self.optimizer_D.load_state_dict(snapshot['optimizer_D'])
        return snapshot

    def create_fake_labels(self):
        """1 for windows made of real frames only (both ends), 0 for every window touching a generated frame
        (environments.py:308-323) -> [K+T+F-disc_t+1]."""
        ones_p = max(0, self.K - self.disc_t + 1)
        ones_f = max(0, self.F - self.disc_t + 1)
        n = self.K + self.T + self.F - self.disc_t + 1
        labels = torch.zeros(n)
        labels[:ones_p] = 1
        if ones_f > 0:
            labels[n - ones_f:] = 1
        return labels

    def _device_fake_labels(self):
        key = (self.K, self.T, self.F)
        if key not in self._fake_labels:
            self._fake_labels[key] = self.create_fake_labels().to(self.device)
        return self._fake_labels[key]

    def _prepare_capture(self):
        self._device_fake_labels()

    def compute_loss_D(self):
        fake = torch.cat([self.preceding_frames, self.gen_output['pred'], self.following_frames], dim=1).detach()
        h = self.discriminator(fake)
        labels = self._device_fake_labels().view(1, -1).expand(fake.size(0), -1)
        self.loss_d_fake = self.loss_d(h, labels)
        real = torch.cat([self.preceding_frames, self.gt_middle_frames, self.following_frames], dim=1).detach()
        h_ = self.discriminator(real)
        self.loss_d_real = self.loss_d(h_, torch.ones_like(h_))
        self.loss_D = self.loss_d_fake + self.loss_d_real

    def optimize_parameters(self):
        super().optimize_parameters()
        self._zero_grad(self.optimizer_D, self._reducer_D)
        self.compute_loss_D()
        self.loss_D.backward()
        self._reducer_D.allreduce_()
        self.optimizer_D.step()

    @staticmethod
    def _time_major_01(x):
        """[B,T,C,H,W] in [-1,1] -> [T*B,C,H,W] in [0,1] (environments.py:363-368)."""
        _, _, c, H, W = x.shape
        return inverse_transform(x.permute(1, 0, 2, 3, 4).contiguous().view(-1, c, H, W))

    def compute_loss_G(self):
        super().compute_loss_G()
        gt = self._time_major_01(self.gt_middle_frames)
        outputs = self._time_major_01(self.gen_output['pred'])
        self.Lp = self.loss_Lp(outputs, gt)
        self.gdl = self.loss_gdl(outputs, gt)
        fake = torch.cat([self.preceding_frames, self.gen_output['pred'], self.following_frames], dim=1)
        # The reference lets this backward pass fill the discriminator's .grad as well and throws those values away
        # (optimizer_D.zero_grad() comes before they are ever read, environments.py:348-355): here the discriminator's
        # parameters are taken out of the graph for this evaluation, which skips a third of its weight-gradient work.
        with _parameters_frozen(self.discriminator):
            h = self.discriminator(fake)
        self.L_GAN = self.loss_d(h, torch.ones_like(h))
        self.loss_G = self.loss_G + self.alpha * (self.Lp + self.gdl) + self.beta * self.L_GAN

    def get_current_errors(self):
        d = super().get_current_errors()
        d.update({'G_Lp': float(self.Lp.item()), 'G_gdl': float(self.gdl.item()),
                  'D_real': float(self.loss_d_real.item()), 'D_fake': float(self.loss_d_fake.item()),
                  'G_GAN': float(self.L_GAN.item())})
        return d

    def train(self):
        super().train()
        self.discriminator.train()


class MCNetTrainingEnvironment(L2GDLDiscTrainingEnvironment):
    """environments.py:400-412."""

    def sample_KTF(self, allow_random_sampling):
        if allow_random_sampling:
            return (self._ktf_rng.randint(2, self.max_K + 1), self._ktf_rng.randint(1, self.max_T + 1),
                    self._ktf_rng.randint(1, self.max_F + 1))
        return self.max_K, self.max_T, self.max_F


class TAITrainingEnvironment(L2GDLDiscTrainingEnvironment):
    """environments.py:415-485: adds alpha (MSE + GDL) on the forward and on the backward intermediate prediction."""

    def sample_KTF(self, allow_random_sampling):
        if allow_random_sampling:
            return (self._ktf_rng.randint(2, self.max_K + 1), self._ktf_rng.randint(1, self.max_T + 1),
                    self._ktf_rng.randint(2, self.max_F + 1))
        return self.max_K, self.max_T, self.max_F

    def compute_loss_G(self):
        super().compute_loss_G()
        gt = self._time_major_01(self.gt_middle_frames)
        out_f = self._time_major_01(self.gen_output['pred_forward'])
        out_b = self._time_major_01(self.gen_output['pred_backward'])
        self.Lp_forward = self.loss_Lp(out_f, gt)
        self.Lp_backward = self.loss_Lp(out_b, gt)
        self.gdl_forward = self.loss_gdl(out_f, gt)
        self.gdl_backward = self.loss_gdl(out_b, gt)
        self.loss_G = self.loss_G + self.alpha * (self.Lp_forward + self.Lp_backward + self.gdl_forward + self.gdl_backward)

    def get_current_errors(self):
        d = super().get_current_errors()
        d.update({'G_Lp_forward': float(self.Lp_forward.item()), 'G_gdl_forward': float(self.gdl_forward.item()),
                  'G_Lp_backward': float(self.Lp_backward.item()), 'G_gdl_backward': float(self.gdl_backward.item())})
        return d
