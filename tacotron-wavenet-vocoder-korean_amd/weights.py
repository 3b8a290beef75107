"""Checkpoint tensors of the WaveNet graph: names, shapes and the canonical blob order.

Names and shapes follow the variable scopes of wavenet/model.py: conv kernels are TensorFlow
(width, in, out); `wavenet/queue/*` state is never part of a restore (generate.py:157)."""
import numpy as np


def tensor_specs(n_layers, R=32, D=32, S=512, Q=256, out_channels=30, scalar_input=True, initial_filter_width=32,
                 use_biases=True, gc_channels=32, gc_cardinality=2, lc_channels=80, upsample_factor=(5, 5, 12)):
    O = out_channels if scalar_input else Q
    specs = [("wavenet/conv1d/kernel", (initial_filter_width, 1, R) if scalar_input else (2, Q, R))]
    if gc_channels and gc_cardinality:      # cardinality None / 0: the caller passes the embedding itself (model.py:199-207)
        specs.append(("wavenet/gc_embedding", (gc_cardinality, gc_channels)))
    for i in range(n_layers):
        p = "wavenet/dilated_stack/layer%d/dilation_layer/" % i
        for nm in ("conv_filter", "conv_gate"):
            specs.append((p + nm + "/kernel", (2, R, D)))
            if use_biases:
                specs.append((p + nm + "/bias", (D,)))
        if gc_channels:
            specs += [(p + "gc_filter/kernel", (1, gc_channels, D)), (p + "gc_gate/kernel", (1, gc_channels, D))]
        if lc_channels:
            specs += [(p + "lc_filter/kernel", (1, lc_channels, D)), (p + "lc_gate/kernel", (1, lc_channels, D))]
        specs.append((p + "dense/kernel", (1, D, R)))
        if use_biases:
            specs.append((p + "dense/bias", (R,)))
        specs.append((p + "skip/kernel", (1, D, S)))
        if use_biases:
            specs.append((p + "skip/bias", (S,)))
    specs.append(("wavenet/conv1d_1/kernel", (1, S, S)))
    if use_biases:
        specs.append(("wavenet/conv1d_1/bias", (S,)))
    specs.append(("wavenet/conv1d_2/kernel", (1, S, O)))
    if use_biases:
        specs.append(("wavenet/conv1d_2/bias", (O,)))
    if lc_channels:
        for i, f in enumerate(upsample_factor):
            specs.append(("wavenet/upsample%d/kernel" % i, (f, 2, 1, 1)))
    return specs


def flatten(specs, tensors):
    """dict name -> array  ==>  canonical float32 blob."""
    parts = []
    for name, shape in specs:
        if name not in tensors:
            raise ValueError("checkpoint tensor %s is missing" % name)
        t = np.asarray(tensors[name], dtype=np.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("tensor %s has shape %s, expected %s" % (name, t.shape, shape))
        parts.append(t.reshape(-1))
    return np.concatenate(parts).astype(np.float32)


def random_tensors(specs, seed=0, scale=0.05):
    """synthetic N(0, scale^2) weights (seed 0, sigma 0.05)."""
    rng = np.random.RandomState(seed)
    return {n: (rng.randn(*shp) * scale).astype(np.float32) for n, shp in specs}
