#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own classes.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What is captured (inputs, weights and outputs, as small .npz files):
  * every MC-Net primitive and kernel-network building block of the reference, imported from
    /root/reference/src and run on CPU: MotionEnc, ContentEnc, CombLayers, Residual, DecCnn (+
    fixed_unpooling), ConvLstmCell, create_basic_conv_block, create_1d_kernel_generator_block,
    the decoder upsample blocks, GDL, bgr2gray(_batched), inverse_transform;
  * whole-model runs of the reference's MCNet.forward and TAIFillInModel.forward (gray/num_block=5/
    ks=51 and color/num_block=4/ks=7) at reduced width (gf_dim=4, kf_dim=2, 32x32), i.e. the reference's own
    control flow: loop counts, list reversal, skip indices, time-ratio injection.

The reference is Python 2.7 / torch 0.3.1 / CUDA-only.  Nothing under /root/reference is modified;
the following in-process accommodations are made, all of them in THIS file:
  * module shims: Queue -> queue, xrange/unicode builtins, empty torchvision and _ext.cunnex modules;
  * Tensor.cuda() is an identity (the reference hard-codes .cuda(), tai.py:72,216);
  * Py2 integer division restored where the reference relies on it: ConvLstmCell's conv padding
    (mcnet.py:278) is reset to (1, 1) and MCNet.get_initial_conv_lstm_state (mcnet.py:378-388) is
    replaced by an H//8 x W//8 zero state;
  * every nn.Upsample instance gets align_corners=True, which is what torch 0.3.1's bilinear
    upsample computed;
  * the sepconv op, for which the reference has no CPU implementation
    (SeparableConvolution.py:48-49), is served by oracle/sepconv_oracle.c -- so whole-model
    fixtures pin everything AROUND the op with the reference's code, and the op itself is pinned
    by the analytic known-answer tests in tests/test_oracle_sepconv.py.
"""
import builtins
import os
import queue
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = '/root/reference'


def install_shims():
    sys.modules['Queue'] = queue
    builtins.xrange = range
    builtins.unicode = str
    tv = types.ModuleType('torchvision')
    tvu = types.ModuleType('torchvision.utils')
    tv.utils = tvu
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.utils'] = tvu
    ext = types.ModuleType('_ext')
    cun = types.ModuleType('_ext.cunnex')
    ext.cunnex = cun
    sys.modules['_ext'] = ext
    sys.modules['_ext.cunnex'] = cun
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self


def seeded_init(module, seed):
    """Deterministic non-trivial weights AND biases (zero biases would hide bias bugs)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if p.dim() > 1:
                fan = p[0].numel() if p.dim() > 1 else p.numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / np.sqrt(max(fan, 1))))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def fix_upsamples(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Upsample):
            m.align_corners = True


def sd_np(module, prefix=''):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def rnd(g, *shape):
    return torch.randn(*shape, generator=g)


def main():
    install_shims()
    from src.models.mcnet import mcnet
    from src.models.tai import tai
    from src.losses.losses import GDL
    from src.util import util
    from oracle import sepconv_oracle

    torch.set_num_threads(4)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    gf = 8
    out = {}

    def put(name, **arrs):
        for k, v in arrs.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().numpy()
            out['%s/%s' % (name, k)] = np.ascontiguousarray(v)

    with torch.no_grad():
        # ---------------- primitives
        m = mcnet.MotionEnc(gf); seeded_init(m, 1)
        x = rnd(g, 2, 1, 32, 32)
        o, res = m(x)
        put('motion_enc', x=x, out=o, res0=res[0], res1=res[1], res2=res[2], **{'w/' + k: v for k, v in sd_np(m).items()})

        m = mcnet.ContentEnc(3, gf); seeded_init(m, 2)
        x = rnd(g, 2, 3, 32, 32)
        o, res = m(x)
        put('content_enc', x=x, out=o, res0=res[0], res1=res[1], res2=res[2], **{'w/' + k: v for k, v in sd_np(m).items()})

        m = mcnet.CombLayers(gf); seeded_init(m, 3)
        a, b = rnd(g, 2, gf * 4, 4, 4), rnd(g, 2, gf * 4, 4, 4)
        put('comb_layers', a=a, b=b, out=m(a, b), **{'w/' + k: v for k, v in sd_np(m).items()})

        m = mcnet.Residual(gf * 4, gf * 2); seeded_init(m, 4)
        a, b = rnd(g, 2, gf * 2, 8, 8), rnd(g, 2, gf * 2, 8, 8)
        put('residual', a=a, b=b, out=m(a, b), **{'w/' + k: v for k, v in sd_np(m).items()})

        m = mcnet.DecCnn(3, gf); seeded_init(m, 5)
        comb = rnd(g, 2, gf * 4, 4, 4)
        r1, r2, r3 = rnd(g, 2, gf, 32, 32), rnd(g, 2, gf * 2, 16, 16), rnd(g, 2, gf * 4, 8, 8)
        put('dec_cnn', comb=comb, r1=r1, r2=r2, r3=r3, out=m(comb, r1, r2, r3),
            unpool=m.fixed_unpooling(comb), **{'w/' + k: v for k, v in sd_np(m).items()})

        m = mcnet.ConvLstmCell(3, gf * 4); seeded_init(m, 6)
        m.conv.padding = (1, 1)          # Py2: (3 - 1) / 2 == 1   (mcnet.py:278)
        inp, st = rnd(g, 2, gf * 4, 4, 4), rnd(g, 2, gf * 8, 4, 4)
        h, ns = m(inp, st)
        put('conv_lstm', inp=inp, state=st, h=h, new_state=ns, **{'w/' + k: v for k, v in sd_np(m).items()})

        # ---------------- kernel-network blocks
        m = tai.create_basic_conv_block(3, 12, 8); seeded_init(m, 7)
        x = rnd(g, 2, 12, 8, 8)
        put('basic_conv_block', x=x, out=m(x), **{'w/' + k: v for k, v in sd_np(m).items()})

        m = tai.create_1d_kernel_generator_block(3, 4, 51); seeded_init(m, 8); fix_upsamples(m)
        x = rnd(g, 2, 8, 8, 8)
        put('kernel_generator_block', x=x, out=m(x), **{'w/' + k: v for k, v in sd_np(m).items()})

        _, ups = tai.create_decoder_blocks(4, 4, 3, 4)
        for i in (0, 3):
            m = ups[i]; seeded_init(m, 9 + i); fix_upsamples(m)
            cin = m[1].weight.shape[1]
            x = rnd(g, 2, cin, 5, 6)          # odd sizes on purpose: align_corners matters
            put('upsample_block_%d' % i, x=x, out=m(x), **{'w/' + k: v for k, v in sd_np(m).items()})

        # ---------------- helpers / losses
        x5 = rnd(g, 2, 3, 3, 8, 8)
        put('util', x5=x5, inv=util.inverse_transform(x5), gray_b=util.bgr2gray_batched(x5),
            gray=util.bgr2gray(x5[:, 0]))
        a, b = rnd(g, 3, 4, 1, 9, 10), rnd(g, 3, 4, 1, 9, 10)
        put('gdl', a=a, b=b, out=GDL()(a, b).reshape(1))

        np.savez_compressed(os.path.join(HERE, 'blocks.npz'), **out)
        print('blocks.npz: %d arrays' % len(out))

        # ---------------- whole-model runs (reference control flow)
        def int_state(self, batch_size, image_size):          # mcnet.py:378-388 under Py2
            return torch.zeros(batch_size, 8 * self.gf_dim, image_size[0] // 8, image_size[1] // 8)

        def cpu_sepconv(inp, v, h, ks=51):                     # SeparableConvolution.py:11-52
            assert inp.shape[2] - ks == v.shape[2] - 1 and inp.shape[3] - ks == v.shape[3] - 1
            return torch.from_numpy(sepconv_oracle.forward(inp.numpy(), v.numpy(), h.numpy(), ks))

        def prep(model):
            fix_upsamples(model)
            gen = model.generator
            gen.conv_lstm_cell.conv.padding = (1, 1)
            gen.get_initial_conv_lstm_state = types.MethodType(int_state, gen)
            if hasattr(model, 'kernelnet'):
                model.kernelnet.separableConvolution = cpu_sepconv

        # MCNet.forward alone (gray), K=4, T=3
        gfm = 4                          # whole-model runs at reduced width keep the fixtures small
        mm = mcnet.MCNetFillInModel(gfm, 1, 3); seeded_init(mm, 20); prep(mm)
        P = torch.tanh(rnd(g, 2, 4, 1, 32, 32))
        pred, dyn, cont, res = mm.generator(4, 3, (P[:, 1:] - P[:, :-1]) / 2, P[:, -1])
        mo = {'P': P.numpy(), 'pred': torch.stack(pred, 1).numpy(), 'dyn': torch.stack(dyn, 1).numpy(),
              'cont': torch.stack(cont, 1).numpy()}
        for t in range(3):
            for i in range(3):
                mo['res_%d_%d' % (t, i)] = res[t][i].numpy()
        mo.update({'w/' + k: v for k, v in sd_np(mm).items()})
        np.savez_compressed(os.path.join(HERE, 'mcnet_gray.npz'), **mo)
        print('mcnet_gray.npz: %d arrays' % len(mo))

        for tag, c_dim, nb, ks, K, Fn, T in (('gray', 1, 5, 51, 3, 4, 3), ('color', 3, 4, 7, 4, 2, 2)):
            tm = tai.TAIFillInModel(gfm, c_dim, 3, ks, num_block=nb, kf_dim=2)
            seeded_init(tm, 30 + c_dim); prep(tm)
            P = torch.tanh(rnd(g, 2, K, c_dim, 32, 32))
            Fo = torch.tanh(rnd(g, 2, Fn, c_dim, 32, 32))
            o = tm(T, P, Fo)
            to = {'P': P.numpy(), 'F': Fo.numpy(), 'T': np.array([T]), 'num_block': np.array([nb]),
                  'c_dim': np.array([c_dim]), 'ks': np.array([ks])}
            to.update({'out/' + k: v.numpy() for k, v in o.items()})
            to.update({'w/' + k: v for k, v in sd_np(tm).items()})
            np.savez_compressed(os.path.join(HERE, 'tai_%s.npz' % tag), **to)
            print('tai_%s.npz: %d arrays, %d params' % (tag, len(to), sum(p.numel() for p in tm.parameters())))


def ablations():
    """bi-TWI, bi-TWA, bi-SA and TW_P_F runs of the reference's classes -> ablations.npz."""
    from src.models.twi import twi
    from src.models.bi_twa import bi_twa
    from src.models.bi_sa import bi_sa
    from src.models.tw_p_f import tw_p_f
    from oracle import sepconv_oracle
    g = torch.Generator().manual_seed(4321)
    out = {}

    def int_state(self, batch_size, image_size):
        return torch.zeros(batch_size, 8 * self.gf_dim, image_size[0] // 8, image_size[1] // 8)

    def cpu_sepconv(inp, v, h, ks=51):
        return torch.from_numpy(sepconv_oracle.forward(inp.numpy(), v.numpy(), h.numpy(), ks))

    def prep_gen(gen):
        gen.conv_lstm_cell.conv.padding = (1, 1)
        gen.get_initial_conv_lstm_state = types.MethodType(int_state, gen)

    with torch.no_grad():
        m = twi.TimeWeightedInterpolationFillInModel(4, 1, 3, 7, num_block=5, kf_dim=2)
        seeded_init(m, 41); fix_upsamples(m); prep_gen(m.mcnet)
        m.interp_net.separableConvolution = cpu_sepconv
        P, Fo = torch.tanh(rnd(g, 2, 3, 1, 32, 32)), torch.tanh(rnd(g, 2, 2, 1, 32, 32))
        o = m(3, P, Fo)
        out.update({'twi/P': P.numpy(), 'twi/F': Fo.numpy()})
        out.update({'twi/out/' + k: v.numpy() for k, v in o.items()})
        out.update({'twi/w/' + k: v for k, v in sd_np(m).items()})
        for tag, cls, c_dim in (('bi_twa', bi_twa.BidirectionalTimeWeightedAverageFillInModel, 1),
                                ('bi_sa', bi_sa.BidirectionalSimpleAverageFillInModel, 3)):
            m = cls(4, c_dim, 3)
            seeded_init(m, 42 + c_dim); prep_gen(m.generator)
            P, Fo = torch.tanh(rnd(g, 2, 3, c_dim, 32, 32)), torch.tanh(rnd(g, 2, 3, c_dim, 32, 32))
            o = m(4, P, Fo)
            out.update({tag + '/P': P.numpy(), tag + '/F': Fo.numpy()})
            out.update({tag + '/out/' + k: v.numpy() for k, v in o.items()})
            out.update({tag + '/w/' + k: v for k, v in sd_np(m).items()})
        m = tw_p_f.TimeWeightedPFFillInModel()
        P, Fo = torch.tanh(rnd(g, 2, 2, 3, 8, 8)), torch.tanh(rnd(g, 2, 2, 3, 8, 8))
        out.update({'tw_p_f/P': P.numpy(), 'tw_p_f/F': Fo.numpy(), 'tw_p_f/out/pred': m(3, P, Fo)['pred'].numpy()})
    np.savez_compressed(os.path.join(HERE, 'ablations.npz'), **out)
    print('ablations.npz: %d arrays' % len(out))


def sn_disc():
    """The reference's spectral-norm discriminator (src/discriminators/SNDiscriminator.py) run on CPU -> sn_disc.npz.

    ``max_singular_value`` / ``_l2normalize`` (:10-33) and ``SNLinear`` (:71-92) run as they are.  ``SNConv2d.__init__``
    (:60-61) passes torch 0.3.1's ten positional arguments to ``_ConvNd.__init__``; torch 2.x added a required
    ``padding_mode`` after them, so that ONE base-class constructor is wrapped here to default it to 'zeros' while the
    reference's modules are constructed (restored afterwards).  ``SNDiscriminator.forward`` (:140-159) needs ``xrange``
    (install_shims).  Every forward of every layer overwrites ``weight.data`` and keeps ``u``, so the fixture records
    weights, u vectors and logits over CONSECUTIVE calls.  Gradients are recorded only for a single-window, single-call
    evaluation: there no later renormalisation touches a weight between its use and the backward pass, so torch 2.x's
    autograd (which reads a leaf parameter's data at backward time) and torch 0.3.1's (which kept the tensor of the
    moment) give the same numbers."""
    from torch.nn.modules import conv as tconv
    orig_init = tconv._ConvNd.__init__

    def init_with_default_padding_mode(self, *args, **kw):
        if len(args) == 10 and 'padding_mode' not in kw:
            args = args + ('zeros',)
        return orig_init(self, *args, **kw)

    tconv._ConvNd.__init__ = init_with_default_padding_mode
    try:
        from src.discriminators import SNDiscriminator as ref
        g = torch.Generator().manual_seed(777)
        out = {}

        def put(name, **arrs):
            for k, v in arrs.items():
                out['%s/%s' % (name, k)] = np.ascontiguousarray(v.detach().numpy() if isinstance(v, torch.Tensor) else v)

        # ---- max_singular_value with a given u, Ip = 1 and 3 (:10-25)
        W = rnd(g, 8, 48) * 0.3
        u = rnd(g, 1, 8)
        for Ip in (1, 3):
            sigma, u_out = ref.max_singular_value(torch.nn.Parameter(W.clone()), u.clone(), Ip=Ip)
            put('msv_ip%d' % Ip, W=W, u=u, sigma=sigma, u_out=u_out)
        v = rnd(g, 1, 13)
        put('l2normalize', v=v, out=ref._l2normalize(v))

        # ---- SNLinear over three consecutive forwards (:84-92)
        lin = ref.SNLinear(48, 1, Ip=1)
        seeded_init(lin, 51)
        lin.u = rnd(g, 1, 1)
        x = rnd(g, 3, 48)
        put('sn_linear', W0=lin.weight.data.clone(), b=lin.bias.data.clone(), u0=lin.u.clone(), x=x)
        for call in range(3):
            y = lin(x)
            put('sn_linear', **{'out%d' % call: y, 'W%d' % (call + 1): lin.weight.data.clone(), 'u%d' % (call + 1): lin.u.clone()})

        # ---- SNConv2d 4x4 stride 2 pad 1 over three consecutive forwards (:60-68)
        cv = ref.SNConv2d(3, 8, 4, stride=2, padding=1, Ip=3)
        seeded_init(cv, 52)
        cv.u = rnd(g, 1, 8)
        x = rnd(g, 2, 3, 16, 16)
        put('sn_conv', W0=cv.weight.data.clone(), b=cv.bias.data.clone(), u0=cv.u.clone(), x=x)
        for call in range(3):
            y = cv(x)
            put('sn_conv', **{'out%d' % call: y, 'W%d' % (call + 1): cv.weight.data.clone(), 'u%d' % (call + 1): cv.u.clone()})

        # ---- SNDiscriminator.forward (:95-159): 5 windows of 3 frames, two consecutive calls on different clips
        for tag, c_dim in (('disc_gray', 1), ('disc_color', 3)):
            D = ref.SNDiscriminator((32, 32), c_dim, 3, 4, 3)
            seeded_init(D, 53 + c_dim)
            for name, m in D.named_modules():
                if hasattr(m, 'Ip'):
                    m.u = rnd(g, 1, m.weight.size(0))
                    put(tag, **{'u0/' + name: m.u.clone()})
            put(tag, **{'w0/' + k: v for k, v in sd_np(D).items()})
            for call in range(2):
                frames = torch.tanh(rnd(g, 2, 7, c_dim, 32, 32))
                logits = D(frames)
                put(tag, **{'frames%d' % call: frames, 'logits%d' % call: logits})
                put(tag, **{'w%d/%s' % (call + 1, k): v for k, v in sd_np(D).items()})
                for name, m in D.named_modules():
                    if hasattr(m, 'Ip'):
                        put(tag, **{'u%d/%s' % (call + 1, name): m.u.clone()})

        # ---- one window, one call, BCE-with-logits against ones, gradients of every parameter
        D = ref.SNDiscriminator((32, 32), 1, 3, 4, 3)
        seeded_init(D, 60)
        for name, m in D.named_modules():
            if hasattr(m, 'Ip'):
                m.u = rnd(g, 1, m.weight.size(0))
                put('disc_grad', **{'u0/' + name: m.u.clone()})
        put('disc_grad', **{'w0/' + k: v for k, v in sd_np(D).items()})
        frames = torch.tanh(rnd(g, 2, 3, 1, 32, 32)).requires_grad_(True)
        logits = D(frames)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
        loss.backward()
        put('disc_grad', frames=frames, logits=logits, loss=loss.reshape(1), grad_frames=frames.grad)
        put('disc_grad', **{'grad/' + k: p.grad for k, p in D.named_parameters()})
        put('disc_grad', **{'w1/' + k: v for k, v in sd_np(D).items()})
        np.savez_compressed(os.path.join(HERE, 'sn_disc.npz'), **out)
        print('sn_disc.npz: %d arrays' % len(out))
    finally:
        tconv._ConvNd.__init__ = orig_init


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'ablations':
        install_shims()
        ablations()
    elif len(sys.argv) > 1 and sys.argv[1] == 'sn_disc':
        install_shims()
        sn_disc()
    else:
        main()
