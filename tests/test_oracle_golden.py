"""Pins oracle/tai_oracle.py against fixtures produced by the reference's own classes
(tests/golden/make_golden.py, run in the build container where /root/reference is mounted)."""
import os

import numpy as np
import pytest
import torch

from oracle import tai_oracle as T

TOL = dict(rtol=1e-5, atol=2e-6)


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def _block(z, name):
    pre = name + '/'
    d = {k[len(pre):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(pre)}
    sd = {k[2:]: v for k, v in d.items() if k.startswith('w/')}
    return d, sd


@pytest.fixture(scope='module')
def blocks(golden_dir):
    return _load(golden_dir, 'blocks.npz')


def _close(a, b):
    np.testing.assert_allclose(a.numpy(), b.numpy(), **TOL)


def test_motion_enc(blocks):
    d, sd = _block(blocks, 'motion_enc')
    out, res = T.motion_enc(sd, '', d['x'])
    _close(out, d['out'])
    for i in range(3):
        _close(res[i], d['res%d' % i])


def test_content_enc(blocks):
    d, sd = _block(blocks, 'content_enc')
    out, res = T.content_enc(sd, '', d['x'])
    _close(out, d['out'])
    for i in range(3):
        _close(res[i], d['res%d' % i])


def test_comb_layers(blocks):
    d, sd = _block(blocks, 'comb_layers')
    _close(T.comb_layers(sd, '', d['a'], d['b']), d['out'])


def test_residual(blocks):
    d, sd = _block(blocks, 'residual')
    _close(T.residual(sd, '', d['a'], d['b']), d['out'])


def test_dec_cnn_and_unpooling(blocks):
    d, sd = _block(blocks, 'dec_cnn')
    _close(T.fixed_unpooling(d['comb']), d['unpool'])
    _close(T.dec_cnn(sd, '', d['comb'], d['r1'], d['r2'], d['r3']), d['out'])


def test_conv_lstm_cell(blocks):
    d, sd = _block(blocks, 'conv_lstm')
    h, ns = T.conv_lstm_cell(sd, '', d['inp'], d['state'])
    _close(h, d['h'])
    _close(ns, d['new_state'])


def test_basic_conv_block(blocks):
    d, sd = _block(blocks, 'basic_conv_block')
    _close(T.basic_conv_block(sd, '', d['x']), d['out'])


def test_kernel_generator_block(blocks):
    d, sd = _block(blocks, 'kernel_generator_block')
    _close(T.kernel_generator_block(sd, '', d['x']), d['out'])


@pytest.mark.parametrize('i', [0, 3])
def test_upsample_block(blocks, i):
    d, sd = _block(blocks, 'upsample_block_%d' % i)
    _close(T.upsample_block(sd, '', d['x']), d['out'])


def test_util_and_gdl(blocks):
    d, _ = _block(blocks, 'util')
    _close(T.inverse_transform(d['x5']), d['inv'])
    _close(T.bgr2gray_batched(d['x5']), d['gray_b'])
    _close(T.bgr2gray(d['x5'][:, 0]), d['gray'])
    d, _ = _block(blocks, 'gdl')
    _close(T.gdl(d['a'], d['b']).reshape(1), d['out'])


def test_mcnet_forward(golden_dir):
    z = _load(golden_dir, 'mcnet_gray.npz')
    sd = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('w/')}
    P = torch.from_numpy(z['P'])
    with torch.no_grad():
        pred, dyn, cont, res = T.mcnet_forward(sd, 'generator.', 1, 4, 3, (P[:, 1:] - P[:, :-1]) / 2, P[:, -1])
    np.testing.assert_allclose(torch.stack(pred, 1).numpy(), z['pred'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(dyn, 1).numpy(), z['dyn'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(cont, 1).numpy(), z['cont'], rtol=1e-4, atol=1e-5)
    for t in range(3):
        for i in range(3):
            np.testing.assert_allclose(res[t][i].numpy(), z['res_%d_%d' % (t, i)], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag', ['gray', 'color'])
def test_tai_forward(golden_dir, tag):
    z = _load(golden_dir, 'tai_%s.npz' % tag)
    sd = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('w/')}
    with torch.no_grad():
        out = T.tai_forward(sd, int(z['c_dim'][0]), int(z['num_block'][0]), int(z['ks'][0]), int(z['T'][0]),
                            torch.from_numpy(z['P']), torch.from_numpy(z['F']))
    for k in ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1', 'interp_net_outputs_2'):
        np.testing.assert_allclose(out[k].numpy(), z['out/' + k], rtol=1e-4, atol=2e-5, err_msg=k)
