"""shared builders for the parity tests: one config -> (oracle dims, oracle blob, product model)."""
import numpy as np


def make_case(O, dilations, scalar_input=True, S=512, Q=256, out_channels=30, ifw=32, use_bias=True, G=32, gc_card=2,
              L=80, up=(5, 5, 12), seed=0, scale=0.05):
    d = O.make_dims(dilations, R=32, D=32, S=S, Q=Q, out_channels=out_channels, scalar_input=scalar_input, ifw=ifw,
                    use_bias=use_bias, G=G, gc_card=gc_card, L=L, up=up)
    tensors = O.random_tensors(d, seed=seed, scale=scale)
    blob = O.blob_from_tensors(d, tensors)
    return d, tensors, blob


def make_model(batch, dilations, tensors, scalar_input=True, S=512, Q=256, out_channels=30, ifw=32, use_bias=True, G=32,
               gc_card=2, L=80, up=(5, 5, 12), workers=None, groups=None, xcd=None, xcd_many=None):
    import twvk_amd  # noqa: F401
    from twvk_amd.wavenet import WaveNetModel
    m = WaveNetModel(batch, dilations, 2, 32, 32, S, quantization_channels=Q, out_channels=out_channels,
                     use_biases=use_bias, scalar_input=scalar_input, initial_filter_width=ifw,
                     global_condition_channels=G or None, global_condition_cardinality=(gc_card or None) if G else None,
                     local_condition_channels=L or None, upsample_factor=list(up) if L else None, train_mode=False)
    if workers:
        m.set_option("workers", workers)
    if groups is not None:
        m.set_option("groups", groups)
    if xcd is not None:
        m.set_option("xcd", xcd)          # 0: the generic kernel even where the XCD-per-stream kernel qualifies
    if xcd_many is not None:
        m.set_option("xcd_many", xcd_many)  # 1: the many-streams XCD kernel (two streams per chain workgroup) also at batch <= 32
    m.load_weights(tensors)
    return m


def mol_uniforms(B, T, nr_mix, seed=2):
    rng = np.random.RandomState(seed)
    r = rng.random_sample((B, T, nr_mix + 1)).astype(np.float32)
    lo, hi = np.float32(1e-5), np.float32(1.0 - 1e-5)
    return (r * (hi - lo) + lo).astype(np.float32)   # tf.random_uniform(minval=1e-5, maxval=1-1e-5), mixture.py:103,110


def first_mismatch(a, b):
    a = np.asarray(a); b = np.asarray(b)
    bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
    return None if bad.size == 0 else tuple(int(v) for v in bad[0])
