"""Plain PyTorch fp32 (CPU, autograd) restatement of the reference's WaveNet training graph -- TEST INFRASTRUCTURE ONLY.

The training step is floating-point work, so its checker is a torch fp32 model of the same op (not the bit-exact C
restatement).  Follows /root/reference:
  wavenet/model.py:247-312 add_loss, :102-111 create_upsample, :66-101 _create_dilation_layer (train_mode=True, 'valid'
  convs, local condition sliced from the FRONT), :112-167 _create_network, wavenet/mixture.py:27-81
  discretized_mix_logistic_loss(num_class=2**16, reduce=False) -> mean, model.py:314-346 Adam (TF formulation) + EMA.
Parity unpinned: TensorFlow is not installed here, so this restatement has not been run against the reference itself."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def log_sum_exp(x, axis=-1):
    m = x.max(dim=axis, keepdim=True)[0]
    return (m + torch.log(torch.sum(torch.exp(x - m), dim=axis, keepdim=True))).squeeze(axis)


def mol_loss(y_hat, y, num_class=2 ** 16, log_scale_min=float(np.log(1e-14))):
    """y_hat (B, T, 3*nr), y (B, T, 1) -> (B, T) negative log-likelihood   [mixture.py:27-81]"""
    nr = y_hat.shape[-1] // 3
    logit_probs = y_hat[:, :, :nr]
    means = y_hat[:, :, nr:2 * nr]
    log_scales = torch.clamp(y_hat[:, :, 2 * nr:3 * nr], min=log_scale_min)
    y = y.expand(-1, -1, nr)
    centered = y - means
    inv_stdv = torch.exp(-log_scales)
    plus_in = inv_stdv * (centered + 1.0 / (num_class - 1))
    cdf_plus = torch.sigmoid(plus_in)
    min_in = inv_stdv * (centered - 1.0 / (num_class - 1))
    cdf_min = torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_stdv * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    log_probs = torch.where(y < -0.999, log_cdf_plus,
                            torch.where(y > 0.999, log_one_minus_cdf_min,
                                        torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)),
                                                    log_pdf_mid - math.log((num_class - 1) / 2))))
    log_probs = log_probs + F.log_softmax(logit_probs, dim=-1)
    return -log_sum_exp(log_probs)


def upsample(lc, kernels, factors):
    """conv2d_transpose(filters=1, kernel=(f,2), strides=(f,1), 'same'): out[t*f+a, m] = K[a,0] in[t,m] + K[a,1] in[t,m-1]"""
    x = lc                                                     # (B, T, L)
    for K, f in zip(kernels, factors):
        K = K.reshape(f, 2)
        xl = F.pad(x, (1, 0))[:, :, :-1]                       # in[t, m-1]
        out = x[:, :, None, :] * K[None, None, :, 0, None] + xl[:, :, None, :] * K[None, None, :, 1, None]
        x = out.reshape(x.shape[0], x.shape[1] * f, x.shape[2])
    return x


def receptive_field(cfg):
    """model.py:31-39"""
    return sum(cfg["dilations"]) + 1 + ((cfg["initial_filter_width"] - 1) if cfg.get("scalar_input", True) else 1)


def network(P, cfg, net_in, U, gc_ids):
    """_create_network(train_mode=True): net_in (B,1,Tn) scalar input, U (B,Tlc,L) upsampled lc -> raw output (B, Tn-rf+1, O)"""
    dil, ifw, ub = cfg["dilations"], cfg["initial_filter_width"], cfg["use_biases"]
    rf = receptive_field(cfg)
    Uc = U.transpose(1, 2)                                     # (B, L, Tlc)
    gc = P["wavenet/gc_embedding"][gc_ids.long()][:, :, None]  # (B, G, 1)

    def conv(x, name, dilation=1, bias=True):                  # TF kernel (W, in, out) -> torch (out, in, W)
        w = P[name + "/kernel"].permute(2, 1, 0)
        b = P[name + "/bias"] if (bias and ub and (name + "/bias") in P) else None
        return F.conv1d(x, w, b, dilation=dilation)

    cur = F.conv1d(net_in, P["wavenet/conv1d/kernel"].permute(2, 1, 0))     # model.py:41-46 (no bias)
    out_w = net_in.shape[2] - rf + 1
    skips = []
    for i, d in enumerate(dil):
        p = "wavenet/dilated_stack/layer%d/dilation_layer/" % i
        f = conv(cur, p + "conv_filter", d)
        g = conv(cur, p + "conv_gate", d)
        f = f + conv(gc, p + "gc_filter", bias=False)
        g = g + conv(gc, p + "gc_gate", bias=False)
        n = f.shape[2]
        f = f + conv(Uc, p + "lc_filter", bias=False)[:, :, :n]            # model.py:79-80 slice from the front
        g = g + conv(Uc, p + "lc_gate", bias=False)[:, :, :n]
        z = torch.tanh(f) * torch.sigmoid(g)
        skips.append(conv(z[:, :, n - out_w:], p + "skip"))
        cur = cur[:, :, cur.shape[2] - n:] + conv(z, p + "dense")
    total = sum(skips)
    c1 = conv(F.relu(total), "wavenet/conv1d_1")
    return conv(F.relu(c1), "wavenet/conv1d_2").transpose(1, 2)             # (B, out_w, 3*nr)


def network_mm(P, cfg, net_in, U, gc_ids):
    """`network` written with matmuls only (a width-2 dilated 'valid' conv is two matmuls on row-shifted slices, a 1x1 conv is one;
    the skip outputs are added up as they appear): device-agnostic and float64-capable on the GPU, where conv1d in float64 is not a
    library path.  Same graph, same result up to the summation order inside a matmul (checked against `network` on the CPU)."""
    dil, ifw, ub = cfg["dilations"], cfg["initial_filter_width"], cfg["use_biases"]
    rf = receptive_field(cfg)
    scalar = cfg.get("scalar_input", True)
    if scalar:
        cur = net_in[:, 0, :].unfold(1, ifw, 1) @ P["wavenet/conv1d/kernel"][:, 0, :]             # (B, Tc, R)
    else:
        x = net_in.transpose(1, 2)                                                                # (B, Tn, Q)
        k = P["wavenet/conv1d/kernel"]
        cur = x[:, :-1] @ k[0] + x[:, 1:] @ k[1]
    out_w = net_in.shape[2] - rf + 1
    gc = P["wavenet/gc_embedding"][gc_ids.long()]                                                 # (B, G)

    def b(name):
        return P[name + "/bias"] if (ub and (name + "/bias") in P) else 0.0

    total = None
    for i, d in enumerate(dil):
        p = "wavenet/dilated_stack/layer%d/dilation_layer/" % i
        n = cur.shape[1] - d
        pre = []
        for nm in ("filter", "gate"):
            w = P[p + "conv_" + nm + "/kernel"]
            v = cur[:, :n] @ w[0] + cur[:, d:] @ w[1] + b(p + "conv_" + nm)
            v = v + (gc @ P[p + "gc_" + nm + "/kernel"][0])[:, None, :]
            v = v + U[:, :n] @ P[p + "lc_" + nm + "/kernel"][0]                                   # model.py:79-80 slice from the front
            pre.append(v)
        z = torch.tanh(pre[0]) * torch.sigmoid(pre[1])
        sk = z[:, n - out_w:] @ P[p + "skip/kernel"][0] + b(p + "skip")
        total = sk if total is None else total + sk
        cur = cur[:, d:] + (z @ P[p + "dense/kernel"][0] + b(p + "dense"))
    c1 = F.relu(total) @ P["wavenet/conv1d_1/kernel"][0] + b("wavenet/conv1d_1")
    return F.relu(c1) @ P["wavenet/conv1d_2/kernel"][0] + b("wavenet/conv1d_2")                  # (B, out_w, O)


def loss_fn(P, cfg, audio, lc, gc_ids, quantized=None, net=None):
    """P: dict TF-name -> torch tensor (TF layouts); audio (B,T); lc (B,T/hop,L); gc_ids (B) -> scalar loss.
    one-hot model (scalar_input False): `quantized` = mu_law_encode(audio) ints (B,T), model.py:257-260"""
    rf = receptive_field(cfg)
    network = net or globals()["network"]
    U = upsample(lc, [P["wavenet/upsample%d/kernel" % i] for i in range(len(cfg["upsample_factor"]))], cfg["upsample_factor"])
    if cfg.get("scalar_input", True):
        net_in = audio[:, None, :-1]                           # model.py:267-269 (B,1,T-1)
        y = network(P, cfg, net_in, U, gc_ids)
        target = audio[:, rf:, None]                                        # model.py:286
        return mol_loss(y, target).mean()
    enc = F.one_hot(quantized.long(), cfg["Q"]).to(audio.dtype)            # (B,T,Q)
    y = network(P, cfg, enc[:, :-1].transpose(1, 2), U, gc_ids)             # (B, out_w, Q)
    tgt = quantized[:, rf:].long()                                          # model.py:286, 293-296
    return F.cross_entropy(y.reshape(-1, cfg["Q"]), tgt.reshape(-1), reduction="mean")


def loss_and_grads(tensors, cfg, audio, lc, gc_ids, dtype=torch.float32, quantized=None, device="cpu", matmul_form=False):
    """device / matmul_form: the full-size configs[3] batch is checked with the matmul form in float64 ON the GPU (torch as the
    checker of a floating-point kernel; conv1d in float64 has no GPU library path)"""
    P = {k: torch.tensor(np.asarray(v), dtype=dtype, device=device, requires_grad=True) for k, v in tensors.items()}
    loss = loss_fn(P, cfg, torch.tensor(audio, dtype=dtype, device=device), torch.tensor(lc, dtype=dtype, device=device),
                   torch.tensor(np.asarray(gc_ids), device=device),
                   None if quantized is None else torch.tensor(np.asarray(quantized), device=device),
                   net=network_mm if matmul_form else None)
    loss.backward()
    return float(loss.item()), {k: (v.grad.cpu().numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in P.items()}


def adam_ema(p, g, m, v, ema, t, lr, b1=0.9, b2=0.999, eps=1e-8, decay=0.9999):
    """tf.train.AdamOptimizer._apply_dense + ExponentialMovingAverage.apply (no num_updates) in float64 numpy"""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    ema = ema - (1 - decay) * (ema - p)
    return p, m, v, ema
