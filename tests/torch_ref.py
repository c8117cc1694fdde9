"""Independent torch-CPU (autograd) formulation of the AE/VAE step, used ONLY by
tests to cross-check the numpy oracle's hand-written backward (the oracle itself
is "parity unpinned": no TF here).  TF-SAME geometry is made explicit:
conv k5 s2 -> F.pad (1,2,1,2); ConvT k5 s2 -> conv_transpose2d(padding=1), crop
the last row/col (SURVEY.md §8a note 2)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
ALPHA = 0.3


def _conv_same(x, w_hwio, b, stride):
    # x NCHW; w HWIO -> OIHW
    k = w_hwio.shape[0]
    h = x.shape[2]
    out = -(-h // stride)
    total = max((out - 1) * stride + k - h, 0)
    pb, pa = total // 2, total - total // 2
    xp = F.pad(x, (pb, pa, pb, pa))
    return F.conv2d(xp, w_hwio.permute(3, 2, 0, 1).contiguous(), b, stride=stride)


def _convT_same(x, w_hwoi, b, stride):
    # TF kernel [kh,kw,Cout,Cin]; torch conv_transpose2d weight [Cin, Cout, kh, kw]
    k = w_hwoi.shape[0]
    h = x.shape[2]
    oh = h * stride
    total = max((h - 1) * stride + k - oh, 0)
    pb = total // 2
    y = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), None, stride=stride, padding=0)
    short = pb + oh - y.shape[2]
    if short > 0:                      # k < stride (k1 s2): TF's output is 2x, the uncovered rows / columns are zero
        y = F.pad(y, (0, short, 0, short))
    y = y[:, :, pb:pb + oh, pb:pb + oh]
    return y + b.view(1, -1, 1, 1)


def forward_loss(arch, spec, params, x_nhwc, eps, masks, inter_res, n_pool):
    """params: dict name -> torch tensor (requires_grad). Returns (loss dict, x_hat NHWC, extras)."""
    rstd = 1.0 / math.sqrt(1.0 + BN_EPS)
    p = params
    a = x_nhwc.permute(0, 3, 1, 2)
    for i in range(n_pool):
        c = _conv_same(a, p[f'Encoder/enc_conv2D_{i}/kernel'], p[f'Encoder/enc_conv2D_{i}/bias'], 2)
        bn = c * (p[f'Encoder/batch_normalization_{i}/gamma'] * rstd).view(1, -1, 1, 1) \
            + p[f'Encoder/batch_normalization_{i}/beta'].view(1, -1, 1, 1)
        a = F.leaky_relu(bn, ALPHA)
    t = _conv_same(a, p['Bottleneck/conv2d/kernel'], p['Bottleneck/conv2d/bias'], 1)
    n = t.shape[0]
    t_nhwc = t.permute(0, 2, 3, 1)
    flat = t_nhwc.reshape(n, -1)
    extras = {}
    if arch == 'VAE':
        mu = flat @ p['Bottleneck/dense_mu/kernel'] + p['Bottleneck/dense_mu/bias']
        ls = flat @ p['Bottleneck/dense_sigma/kernel'] + p['Bottleneck/dense_sigma/bias']
        if 'mu' in masks:
            mu = mu * masks['mu']
        if 'sigma' in masks:
            ls = ls * masks['sigma']
        sg = torch.exp(ls)
        z = mu + eps * sg
        extras.update(z_mu=mu, z_log_sigma=ls, z_sigma=sg)
    else:
        z = flat @ p['Bottleneck/dense_z/kernel'] + p['Bottleneck/dense_z/bias']
        if 'z' in masks:
            z = z * masks['z']
        extras['z'] = z
    d = z @ p['Bottleneck/dense_dec/kernel'] + p['Bottleneck/dense_dec/bias']
    if arch == 'VAE' and 'dec' in masks:
        d = d * masks['dec']
    d4 = d.reshape(t_nhwc.shape).permute(0, 3, 1, 2)
    c = _conv_same(d4, p['Bottleneck/conv2d_1/kernel'], p['Bottleneck/conv2d_1/bias'], 1)
    bn = c * (p['Decoder/batch_normalization/gamma'] * rstd).view(1, -1, 1, 1) \
        + p['Decoder/batch_normalization/beta'].view(1, -1, 1, 1)
    a = F.relu(bn)
    for i in range(n_pool):
        c = _convT_same(a, p[f'Decoder/dec_Conv2DT_{i}/kernel'], p[f'Decoder/dec_Conv2DT_{i}/bias'], 2)
        bn = c * (p[f'Decoder/batch_normalization_{i + 1}/gamma'] * rstd).view(1, -1, 1, 1) \
            + p[f'Decoder/batch_normalization_{i + 1}/beta'].view(1, -1, 1, 1)
        a = F.leaky_relu(bn, ALPHA)
    xh = _conv_same(a, p['Decoder/dec_Conv2D_final/kernel'], p['Decoder/dec_Conv2D_final/bias'], 1)
    xh_nhwc = xh.permute(0, 2, 3, 1)
    l1 = (xh_nhwc - x_nhwc).abs()
    rec = l1.reshape(n, -1).sum(dim=1)
    losses = {'reconstructionLoss': rec.mean()}
    if arch == 'VAE':
        # trainers/VAE.py:38 literally: 0.5*sum(mu^2 + sigma^2 - log(sigma^2) - 1)
        kl = 0.5 * (mu ** 2 + sg ** 2 - torch.log(sg ** 2) - 1).sum(dim=1)
        losses['kl'] = kl.mean()
        losses['loss'] = (rec + kl).mean()
    else:
        losses['loss'] = losses['reconstructionLoss']
    return losses, xh_nhwc, extras


def to_torch(params_np, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in params_np.items()}


def cevae_losses(params, x, x_ce, eps, masks, n_pool):
    """ceVAE graph written the way the reference builds it (context_encoder_variational_autoencoder.py:9-59): one set
    of layers applied to x and to x_ce inside ONE autograd graph; losses as trainers/ceVAE.py:38-51.  Returns
    (losses incl. 'anomaly', x_hat, x_hat_ce); parameter gradients come from losses['loss'].backward()."""
    rstd = 1.0 / math.sqrt(1.0 + BN_EPS)
    p = params

    def bn(c, scope):
        return c * (p[scope + '/gamma'] * rstd).view(1, -1, 1, 1) + p[scope + '/beta'].view(1, -1, 1, 1)

    def encoder(img):
        a = img.permute(0, 3, 1, 2)
        for i in range(n_pool):
            c = _conv_same(a, p[f'Encoder/enc_conv2D_{i}/kernel'], p[f'Encoder/enc_conv2D_{i}/bias'], 2)
            a = F.leaky_relu(bn(c, f'Encoder/batch_normalization_{i}'), ALPHA)
        t = _conv_same(a, p['Bottleneck/conv2d/kernel'], p['Bottleneck/conv2d/bias'], 1).permute(0, 2, 3, 1)
        return t.reshape(t.shape[0], -1), t.shape

    def decoder(z, mask, tshape):
        d = z @ p['Bottleneck/dense_2/kernel'] + p['Bottleneck/dense_2/bias']
        if mask is not None:
            d = d * mask
        a = _conv_same(d.reshape(tshape).permute(0, 3, 1, 2), p['Bottleneck/conv2d_1/kernel'],
                       p['Bottleneck/conv2d_1/bias'], 1)
        a = F.relu(bn(a, 'Decoder/batch_normalization'))
        for i in range(n_pool):
            c = _convT_same(a, p[f'Decoder/dec_Conv2DT_{i}/kernel'], p[f'Decoder/dec_Conv2DT_{i}/bias'], 2)
            a = F.leaky_relu(bn(c, f'Decoder/batch_normalization_{i + 1}'), ALPHA)
        xh = _conv_same(a, p['Decoder/dec_Conv2D_final/kernel'], p['Decoder/dec_Conv2D_final/bias'], 1)
        return xh.permute(0, 2, 3, 1)

    x = x.clone().requires_grad_(True)
    flat, tshape = encoder(x)
    flat_ce, _ = encoder(x_ce)
    mu = flat @ p['Bottleneck/dense/kernel'] + p['Bottleneck/dense/bias']
    mu_ce = flat_ce @ p['Bottleneck/dense/kernel'] + p['Bottleneck/dense/bias']
    ls = flat @ p['Bottleneck/dense_1/kernel'] + p['Bottleneck/dense_1/bias']
    if 'mu' in masks:
        mu = mu * masks['mu']
    if 'mu_ce' in masks:
        mu_ce = mu_ce * masks['mu_ce']
    if 'sigma' in masks:
        ls = ls * masks['sigma']
    sg = torch.exp(ls)
    x_hat = decoder(mu + eps * sg, masks.get('dec'), tshape)
    x_hat_ce = decoder(mu_ce, masks.get('dec_ce'), tshape)
    n = x.shape[0]
    l1v, l1c = (x - x_hat).abs(), (x_ce - x_hat_ce).abs()
    rv, rc = l1v.reshape(n, -1).sum(dim=1), l1c.reshape(n, -1).sum(dim=1)
    kl = 0.5 * (mu ** 2 + sg ** 2 - torch.log(sg ** 2) - 1).sum(dim=1)
    L = {'L1_vae': l1v, 'L1_ce': l1c, 'L1': 0.5 * (l1v + l1c), 'Rec_ce': rc.mean(), 'Rec_vae': rv.mean(),
         'reconstructionLoss': 0.5 * (rv + rc).mean(), 'kl': kl.mean(), 'loss': (rv + kl + rc).mean(),
         'loss_vae': (rv + kl).mean()}
    gx, = torch.autograd.grad(L['loss_vae'], x, retain_graph=True)
    L['anomaly'] = l1v.detach() * gx.abs()
    return L, x_hat, x_hat_ce


def gmvae_losses(params, bn_names, x, e_w, e_z, n_pool, dim_c, dim_z, c_lambda, tv_lambda):
    """Spatial GMVAE graph + losses written the way the reference builds them
    (models/gaussian_mixture_variational_autoencoder_spatial.py:9-65, trainers/GMVAE_spatial.py:61-92), in autograd."""
    rstd = 1.0 / math.sqrt(1.0 + BN_EPS)
    p = params

    def bn(c, scope):
        return c * (p[scope + '/gamma'] * rstd).view(1, -1, 1, 1) + p[scope + '/beta'].view(1, -1, 1, 1)

    def conv1(t, name):
        return _conv_same(t, p[name + '/kernel'], p[name + '/bias'], 1)

    x = x.clone().requires_grad_(True)
    a = x.permute(0, 3, 1, 2)
    for i in range(n_pool):
        a = F.leaky_relu(bn(_conv_same(a, p[f'enc_conv2D_{i}/kernel'], p[f'enc_conv2D_{i}/bias'], 2), bn_names[i]), ALPHA)
    h = a
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    w_mu, w_ls = nhwc(conv1(h, 'q_wz_x/w_mu')), nhwc(conv1(h, 'q_wz_x/w_log_sigma'))
    z_mu, z_ls = nhwc(conv1(h, 'q_wz_x/z_mu')), nhwc(conv1(h, 'q_wz_x/z_log_sigma'))
    w_s = w_mu + e_w * torch.exp(0.5 * w_ls)
    z_s = z_mu + e_z * torch.exp(0.5 * z_ls)
    mid = F.relu(conv1(nchw(w_s), 'p_z_wc/1x1convlayer'))
    n, hh, ww = z_mu.shape[:3]
    M = nhwc(conv1(mid, 'p_z_wc/z_wc_mu')).reshape(n, hh, ww, dim_z, dim_c)
    Lq = (nhwc(conv1(mid, 'p_z_wc/z_wc_log_sigma')) + p['Variable']).reshape(n, hh, ww, dim_z, dim_c)
    a = F.relu(bn(h, bn_names[n_pool]))
    for i in range(n_pool):
        c = _convT_same(a, p[f'dec_Conv2DT_{i}/kernel'], p[f'dec_Conv2DT_{i}/bias'], 2)
        a = F.leaky_relu(bn(c, bn_names[n_pool + 1 + i]), ALPHA)
    xh = nhwc(_conv_same(a, p['dec_Conv2D_final/kernel'], p['dec_Conv2D_final/bias'], 1))
    z_t = z_s.unsqueeze(-1).expand(-1, -1, -1, -1, dim_c)
    loglh = -0.5 * ((z_t - M) ** 2 * torch.exp(Lq)) - Lq + math.log(math.pi)
    logit = loglh.sum(dim=3)
    pc = torch.softmax(logit, dim=-1)
    L = {}
    l1 = (x - xh).abs()
    L['mean_p_loss'] = l1.reshape(n, -1).sum(dim=1).mean()
    zm = z_mu.unsqueeze(-1).expand(-1, -1, -1, -1, dim_c)
    zl = z_ls.unsqueeze(-1).expand(-1, -1, -1, -1, dim_c)
    d_var = (torch.exp(zl) + (zm - M) ** 2) * (torch.exp(Lq) + 1e-6)
    kl = (d_var - (Lq + zl) - 1) * 0.5
    con = torch.matmul(kl, pc.unsqueeze(-1)).squeeze(-1).sum(dim=(1, 2, 3))
    L['conditional_prior_loss'] = con.mean()
    L['w_prior_loss'] = (0.5 * (w_mu ** 2 + torch.exp(w_ls) - w_ls - 1).sum(dim=(1, 2, 3))).mean()
    closs1 = (pc * torch.log(pc * dim_c + 1e-8)).sum(dim=3)
    L['c_prior_loss'] = torch.maximum(closs1, torch.full_like(closs1, c_lambda)).sum(dim=(1, 2)).mean()
    L['loss'] = L['mean_p_loss'] + L['conditional_prior_loss'] + L['w_prior_loss'] + L['c_prior_loss']
    r = x - xh
    tv = (r[:, 1:] - r[:, :-1]).abs().sum(dim=(1, 2, 3)) + (r[:, :, 1:] - r[:, :, :-1]).abs().sum(dim=(1, 2, 3))
    L['restore'] = tv_lambda * tv
    L['grads'], = torch.autograd.grad((L['loss'] + L['restore']).sum(), x, retain_graph=True)      # tf.gradients sums the [n]-shaped ys
    L['dx_loss'], = torch.autograd.grad(L['loss'], x, retain_graph=True)
    return L, xh, {'pc': pc, 'z_wc_mus': M, 'z_wc_log_sigma_invs': Lq, 'w_sampled': w_s, 'z_sampled': z_s}


# ---------------------------------------------------------------------------------------------------------------
# f-AnoGAN (unified graph): models/fanogan.py:11-84 + the loss graph of trainers/fAnoGAN.py:50-66, written with autograd
# (torch.autograd.grad(create_graph=True) plays tf.gradients(d_hat, x_hat) inside the penalty).
# ---------------------------------------------------------------------------------------------------------------
LN_EPS = 1e-3


def _ln_hw(c, gamma, beta):
    # c NCHW; keras LayerNormalization([1, 2]) of the NHWC tensor = statistics over (H, W), gamma/beta [H, W]
    mu = c.mean(dim=(2, 3), keepdim=True)
    var = ((c - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (c - mu) / torch.sqrt(var + LN_EPS) * gamma[None, None] + beta[None, None]


def fanogan_graph(P, x_nhwc, z, alpha, n_pool, inter_res, scale=10.0, kappa=1.0, mask_z=None, mask_g=None, mask_g_enc=None):
    """P: name -> tensor.  Returns the dict of graph outputs / losses (torch tensors, NHWC where images)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    n = x.shape[0]
    ln_names = sorted({k.rsplit('/', 1)[0] for k in P if 'layer_normalization' in k},
                      key=lambda s: int(s.rsplit('_', 1)[1]) if s.rsplit('_', 1)[1].isdigit() else 0)
    ln_g = [k for k in ln_names if k.startswith('Generator/')]
    ln_d = [k for k in ln_names if k.startswith('Discriminator/')]

    def encoder(a):
        for i in range(n_pool):
            a = _conv_same(a, P['Encoder/enc_conv2D_%d/kernel' % i], P['Encoder/enc_conv2D_%d/bias' % i], 2)
            bn = 'Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)
            a = a * (P[bn + '/gamma'] / math.sqrt(1.0 + BN_EPS)).view(1, -1, 1, 1) + P[bn + '/beta'].view(1, -1, 1, 1)
            a = F.leaky_relu(a, ALPHA)
        t = _conv_same(a, P['Encoder/conv2d/kernel'], P['Encoder/conv2d/bias'], 1)
        flat = t.permute(0, 2, 3, 1).reshape(n, -1)
        zr = flat @ P['Encoder/dense/kernel'] + P['Encoder/dense/bias']
        if mask_z is not None:
            zr = zr * mask_z
        return torch.tanh(zr)

    def generator(zz, mask):
        dv = zz @ P['Generator/dense/kernel'] + P['Generator/dense/bias']
        if mask is not None:
            dv = dv * mask
        a = dv.reshape(n, inter_res, inter_res, -1).permute(0, 3, 1, 2)
        a = _conv_same(a, P['Generator/conv2d_1/kernel'], P['Generator/conv2d_1/bias'], 1)
        a = F.relu(_ln_hw(a, P[ln_g[0] + '/gamma'], P[ln_g[0] + '/beta']))
        for i in range(n_pool):
            a = _convT_same(a, P['Generator/dec_Conv2DT_%d/kernel' % i], P['Generator/dec_Conv2DT_%d/bias' % i], 2)
            a = F.leaky_relu(_ln_hw(a, P[ln_g[i + 1] + '/gamma'], P[ln_g[i + 1] + '/beta']), ALPHA)
        return torch.sigmoid(_conv_same(a, P['Generator/dec_Conv2D_final/kernel'], P['Generator/dec_Conv2D_final/bias'], 1))

    def critic(a):
        for i in range(n_pool):
            a = _conv_same(a, P['Discriminator/enc_conv2D_%d/kernel' % i], P['Discriminator/enc_conv2D_%d/bias' % i], 2)
            a = F.leaky_relu(_ln_hw(a, P[ln_d[i] + '/gamma'], P[ln_d[i] + '/beta']), ALPHA)
        feat = a.permute(0, 2, 3, 1)                                   # NHWC
        return feat, feat @ P['Discriminator/dense/kernel'] + P['Discriminator/dense/bias']

    o = {}
    o['z_enc'] = z_enc = encoder(x)
    o['x_enc'] = x_enc = generator(z_enc, mask_g_enc)
    o['x_'] = x_ = generator(z, mask_g)
    o['d_fake_features'], o['d_'] = critic(x_)
    o['d_features'], o['d'] = critic(x)
    x_hat = x + alpha.view(n, 1, 1, 1) * (x_ - x)
    o['d_hat_features'], o['d_hat'] = critic(x_hat)
    o['d_enc_features'], o['d_enc'] = critic(x_enc)
    o['disc_real'] = o['d'].mean()
    o['disc_fake'] = o['d_'].mean()
    o['gen_loss'] = -o['disc_fake']
    ddx = torch.autograd.grad(o['d_hat'].sum(), x_hat, create_graph=True)[0].permute(0, 2, 3, 1)   # NHWC
    slopes = torch.sqrt((ddx ** 2).sum(dim=1))
    o['penalty'] = ((slopes - 1.0) ** 2).mean() * scale
    o['disc_loss'] = o['disc_fake'] - o['disc_real'] + o['penalty']
    xe, xx = x_enc.permute(0, 2, 3, 1), x_nhwc
    o['loss_img'] = ((xx - xe) ** 2).mean(dim=(1, 2, 3)).mean()
    o['loss_fts'] = ((o['d_enc_features'] - o['d_features']) ** 2).mean(dim=(1, 2, 3)).mean()
    o['enc_loss'] = o['loss_img'] + kappa * o['loss_fts']
    o['reconstructionLoss'] = (xx - xe).abs().sum(dim=(1, 2, 3)).mean()
    o['x_enc'] = xe
    o['x_'] = x_.permute(0, 2, 3, 1)
    o['ddx'] = ddx
    return o


# ---------------------------------------------------------------------------------------------------------------
# f-AnoGAN, ResNet graph (models/fanogan_schlegl.py:11-161) with autograd; layer names come from the oracle's table.
# ---------------------------------------------------------------------------------------------------------------
def fanogan_schlegl_graph(P, blocks_g, blocks_d, names, x_nhwc, z, alpha, inter_res, scale=10.0, kappa=1.0):
    x = x_nhwc.permute(0, 3, 1, 2)
    n = x.shape[0]

    def conv(a, name, stride):
        return _conv_same(a, P[name + '/kernel'], P[name + '/bias'], stride)

    def convT(a, name, stride):
        return _convT_same(a, P[name + '/kernel'], P[name + '/bias'], stride)

    def ln(a, name):
        return _ln_hw(a, P[name + '/gamma'], P[name + '/beta'])

    def encoder(a):
        for i in range(3):
            a = conv(a, 'Encoder/enc_conv2D_%d' % i, 2)
            bn = 'Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)
            a = F.leaky_relu(a * (P[bn + '/gamma'] / math.sqrt(1.0 + BN_EPS)).view(1, -1, 1, 1) + P[bn + '/beta'].view(1, -1, 1, 1), ALPHA)
        return torch.tanh(a.permute(0, 2, 3, 1).reshape(n, -1) @ P['Encoder/dense/kernel'] + P['Encoder/dense/bias'])

    def generator(zz):
        out = (zz @ P['Generator/dense/kernel'] + P['Generator/dense/bias']).reshape(n, inter_res, inter_res, -1).permute(0, 3, 1, 2)
        for b in blocks_g:
            t = convT(F.relu(ln(conv(F.relu(ln(out, b.n['ln1'])), b.n['conv1'], 1), b.n['ln2'])), b.n['conv2'], b.stride)
            out = t + (out if b.n['short'] is None else convT(out, b.n['short'], 2))
        return torch.tanh(conv(F.relu(ln(out, names['gen_ln'])), names['gen_final'], 1))

    def critic(a):
        out = conv(a, names['dis_conv'], 1)
        for b in blocks_d:
            t = conv(F.relu(ln(conv(F.relu(ln(out, b.n['ln1'])), b.n['conv1'], 1), b.n['ln2'])), b.n['conv2'], b.stride)
            out = t + (out if b.n['short'] is None else F.avg_pool2d(conv(out, b.n['short'], 1), 2))
        feat = out.permute(0, 2, 3, 1)
        return feat, feat @ P['Discriminator/dense/kernel'] + P['Discriminator/dense/bias']

    o = {}
    o['z_enc'] = z_enc = encoder(x)
    x_enc = generator(z_enc)
    x_ = generator(z)
    o['d_fake_features'], o['d_'] = critic(x_)
    o['d_features'], o['d'] = critic(x)
    x_hat = x + alpha.view(n, 1, 1, 1) * (x_ - x)
    _, o['d_hat'] = critic(x_hat)
    o['d_enc_features'], _ = critic(x_enc)
    o['disc_real'], o['disc_fake'] = o['d'].mean(), o['d_'].mean()
    o['gen_loss'] = -o['disc_fake']
    ddx = torch.autograd.grad(o['d_hat'].sum(), x_hat, create_graph=True)[0].permute(0, 2, 3, 1)
    o['penalty'] = ((torch.sqrt((ddx ** 2).sum(dim=1)) - 1.0) ** 2).mean() * scale
    o['disc_loss'] = o['disc_fake'] - o['disc_real'] + o['penalty']
    xe = x_enc.permute(0, 2, 3, 1)
    o['loss_img'] = ((x_nhwc - xe) ** 2).mean()
    o['loss_fts'] = ((o['d_enc_features'] - o['d_features']) ** 2).mean()
    o['enc_loss'] = o['loss_img'] + kappa * o['loss_fts']
    o['reconstructionLoss'] = (x_nhwc - xe).abs().sum(dim=(1, 2, 3)).mean()
    o['x_enc'], o['x_'] = xe, x_.permute(0, 2, 3, 1)
    return o
