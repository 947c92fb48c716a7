"""CPU restatement of ONE bi-TAI training step (reference train.py:102-119) -- TEST INFRASTRUCTURE ONLY.

Follows, as flat functions over state dicts:
  * the spectral-norm sliding-window discriminator, src/discriminators/SNDiscriminator.py:10-33 (power iteration),
    :63-68 / :87-92 (every forward overwrites ``weight.data <- weight.data / sigma`` and keeps ``u``), :140-159 (windows);
  * the generator loss of TAITrainingEnvironment, src/environments/environments.py:358-379 and :429-453:
        loss_G = alpha (MSE + GDL)(pred) + beta BCEWithLogits(D(cat[P, pred, F]), 1)
                 + alpha (MSE + GDL)(pred_forward) + alpha (MSE + GDL)(pred_backward),
    on tensors mapped to [0, 1] and regrouped time-major [T*B, C, H, W];
  * the discriminator loss, :326-345 with the window labels of :308-323;
  * GDL, src/losses/losses.py:17-44 (oracle/tai_oracle.gdl);
  * the generator itself with autograd through the C sepconv oracle's backward kernels (tai_oracle.SepconvFunction).
The order of discriminator evaluations is the reference's (optimize_parameters, :348-355): D(fake) inside the G loss,
then D(fake.detach()) and D(real) for the D loss; every evaluation renormalises every layer once per window, so the
discriminator's weights and ``u`` vectors are threaded through as mutable state exactly as the modules mutate them.

No optimiser here: the check is on loss terms and gradients (Adam's first step is lr * sign(g), which turns fp32 noise on
near-zero gradient entries into +-lr weight differences -- not a meaningful comparison).
"""
import torch
import torch.nn.functional as F

from . import tai_oracle

SN_CONV_KEYS = ('conv_layers.0', 'conv_layers.2', 'conv_layers.4', 'conv_layers.6')


def _l2normalize(v, eps=1e-12):
    """SNDiscriminator.py:28-33"""
    return v / (((v ** 2).sum()) ** 0.5 + eps)


def max_singular_value(W, u, Ip):
    """SNDiscriminator.py:10-25 (u must be given: the reference draws it N(0,1) on first use, the tests fix it)."""
    _u = u
    for _ in range(Ip):
        _v = _l2normalize(torch.matmul(_u, W), eps=1e-12)
        _u = _l2normalize(torch.matmul(_v, W.t()), eps=1e-12)
    sigma = torch.matmul(torch.matmul(_v, W.t()), _u.t())
    return sigma, _u


class DiscriminatorState(object):
    """Weights, biases and u vectors of the SN discriminator, mutated by every forward as the reference's modules are."""

    def __init__(self, state_dict, u, Ip, window_size):
        self.sd = {k: v.detach().clone() for k, v in state_dict.items()}
        self.u = {k: v.detach().clone() for k, v in u.items()}
        self.Ip, self.window_size = Ip, window_size
        self.uses = []          # (key, tensor used in a forward) in call order: gradients are summed per key

    def _renormalised(self, key, Ip, track):
        W = self.sd[key + '.weight']
        sigma, u = max_singular_value(W.view(W.size(0), -1), self.u[key], Ip)
        self.u[key] = u
        W = (W / sigma).detach()                                        # weight.data = weight.data / sigma  (:67, :91)
        self.sd[key + '.weight'] = W
        if track:
            W = W.clone().requires_grad_(True)
            self.uses.append((key + '.weight', W))
        return W

    def forward(self, frames, track=False, bias_leaves=None, masks=None):
        """SNDiscriminator.forward, :140-159: frames [B, T, C, H, W] -> logits [B, T - window + 1].

        ``masks`` ({(window, layer key): bool [B, Co, H', W']}, optional): the side of LeakyReLU's kink each pre-activation is
        taken to lie on, given from outside instead of read from this run's own rounding.  The gradient is discontinuous there:
        ONE pre-activation of conv_layers.0 that rounds to the other side of zero moves that layer's weight gradient by 3e-3 of its
        maximum (tests/test_train_oracle_cpu.py pins this with an fp64 run), so a gradient comparison between two fp32
        implementations is only well-posed once both differentiate the same piecewise-linear function.  Values are unaffected to
        rounding (the elements concerned are at rounding distance from zero)."""
        B, T, C, H, W = frames.shape
        outs = []
        for t0 in range(T - self.window_size + 1):
            x = frames[:, t0:t0 + self.window_size].contiguous().view(B, self.window_size * C, H, W)
            for key in SN_CONV_KEYS:
                w = self._renormalised(key, self.Ip, track)
                b = bias_leaves[key + '.bias'] if bias_leaves is not None else self.sd[key + '.bias']
                y = F.conv2d(x, w, b, stride=2, padding=1)
                x = F.leaky_relu(y, 0.2) if masks is None else torch.where(masks[(t0, key)], y, 0.2 * y)
            w = self._renormalised('linear_layer', 1, track)                     # SNLinear(..., Ip=1), :136
            b = bias_leaves['linear_layer.bias'] if bias_leaves is not None else self.sd['linear_layer.bias']
            outs.append(F.linear(x.reshape(B, -1), w, b))
        return torch.cat(outs, dim=1)


def fake_labels(K, T, Fn, disc_t):
    """create_fake_labels, environments.py:308-323."""
    ones_p, ones_f = max(0, K - disc_t + 1), max(0, Fn - disc_t + 1)
    n = K + T + Fn - disc_t + 1
    parts = []
    if ones_p > 0:
        parts.append(torch.ones(ones_p))
    parts.append(torch.zeros(n - ones_p - ones_f))
    if ones_f > 0:
        parts.append(torch.ones(ones_f))
    return torch.cat(parts)


def _time_major_01(x):
    """environments.py:363-368: [B,T,C,H,W] in [-1,1] -> [T*B,C,H,W] in [0,1]."""
    _, _, c, H, W = x.shape
    return tai_oracle.inverse_transform(x.permute(1, 0, 2, 3, 4).contiguous().view(-1, c, H, W))


def generator_leg(gen_sd, disc, c_dim, num_block, ks, P, GT, Fo, alpha, beta, grad_keys):
    """The G half of the step (forward_train + the generator loss and its gradients; environments.py:173-176, :358-379,
    :429-453).  ``disc`` is mutated by the one discriminator evaluation inside the loss.  Returns (loss terms as tensors,
    {generator key: grad}, outputs dict, fake = cat[P, pred, F] detached)."""
    K, T, Fn = P.shape[1], GT.shape[1], Fo.shape[1]
    sd = {k: v.detach().clone() for k, v in gen_sd.items()}
    for k in grad_keys:
        sd[k].requires_grad_(True)
    out = tai_oracle.tai_forward(sd, c_dim, num_block, ks, T, P, Fo)              # forward_train, :173-176

    # ---- generator loss (environments.py:358-379, :429-453)
    gt = _time_major_01(GT)
    terms = {}
    for tag, key in (('', 'pred'), ('_forward', 'pred_forward'), ('_backward', 'pred_backward')):
        o = _time_major_01(out[key])
        terms['G_Lp' + tag] = F.mse_loss(o, gt)
        terms['G_gdl' + tag] = tai_oracle.gdl(o, gt)
    fake = torch.cat([P, out['pred'], Fo], dim=1)
    h = disc.forward(fake)
    terms['G_GAN'] = F.binary_cross_entropy_with_logits(h, torch.ones_like(h))
    loss_G = alpha * (terms['G_Lp'] + terms['G_gdl']) + beta * terms['G_GAN'] + alpha * (
        terms['G_Lp_forward'] + terms['G_Lp_backward'] + terms['G_gdl_forward'] + terms['G_gdl_backward'])
    terms['G_loss'] = loss_G
    g_grads = dict(zip(grad_keys, torch.autograd.grad(loss_G, [sd[k] for k in grad_keys])))
    return {k: v.detach() for k, v in terms.items()}, g_grads, {k: v.detach() for k, v in out.items()}, fake.detach()


def discriminator_leg(disc, fake, P, GT, Fo, masks_fake=None, masks_real=None):
    """The D half (environments.py:326-345), after the G half in the reference's order (:348-355): D(fake.detach()) against the
    window labels of :308-323, D(real) against ones.  ``masks_*``: see DiscriminatorState.forward.  Returns (loss terms,
    {discriminator key: grad} for every discriminator parameter)."""
    K, T, Fn = P.shape[1], GT.shape[1], Fo.shape[1]
    terms = {}
    disc.uses = []
    bias_leaves = {k: v.clone().requires_grad_(True) for k, v in disc.sd.items() if k.endswith('.bias')}
    hf = disc.forward(fake.detach(), track=True, bias_leaves=bias_leaves, masks=masks_fake)
    labels = fake_labels(K, T, Fn, disc.window_size).view(1, -1).expand(fake.size(0), -1).to(hf.dtype)
    terms['D_fake'] = F.binary_cross_entropy_with_logits(hf, labels)
    real = torch.cat([P, GT, Fo], dim=1)
    hr = disc.forward(real, track=True, bias_leaves=bias_leaves, masks=masks_real)
    terms['D_real'] = F.binary_cross_entropy_with_logits(hr, torch.ones_like(hr))
    loss_D = terms['D_fake'] + terms['D_real']
    leaves = [w for _, w in disc.uses] + list(bias_leaves.values())
    grads = torch.autograd.grad(loss_D, leaves)
    d_grads = {}
    for (key, _), g in zip(disc.uses, grads[:len(disc.uses)]):          # the same Parameter accumulates every window's grad
        d_grads[key] = d_grads.get(key, 0) + g
    for key, g in zip(bias_leaves, grads[len(disc.uses):]):
        d_grads[key] = g
    return {k: v.detach() for k, v in terms.items()}, d_grads


def training_step(gen_sd, disc, c_dim, num_block, ks, P, GT, Fo, alpha, beta, grad_keys, masks_fake=None, masks_real=None):
    """One G-then-D step without the optimiser updates.  ``gen_sd``: generator state dict (reference schema);
    ``disc``: DiscriminatorState (mutated).  Returns (losses dict, {generator key: grad} for ``grad_keys``,
    {discriminator key: grad} for every discriminator parameter, outputs dict)."""
    terms, g_grads, out, fake = generator_leg(gen_sd, disc, c_dim, num_block, ks, P, GT, Fo, alpha, beta, grad_keys)
    d_terms, d_grads = discriminator_leg(disc, fake, P, GT, Fo, masks_fake, masks_real)
    terms.update(d_terms)
    losses = {k: float(v) for k, v in terms.items()}
    return losses, g_grads, d_grads, out
