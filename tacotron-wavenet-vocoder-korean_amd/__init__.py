"""MI355X-native WaveNet-vocoder hot path of hccho2/Tacotron-Wavenet-Vocoder-Korean.

The package directory name carries hyphens (it is fixed by the project layout); import it through the
top-level shim `twvk_amd` (repo root), which registers this directory as the package `twvk_amd`."""
from . import _lib                                   # noqa: F401
from .hparams import hparams, HParams, default_hparams, load_hparams, save_hparams   # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `build()` works in a bare environment
    if name in ("WaveNetModel",):
        from .wavenet import WaveNetModel
        return WaveNetModel
    if name in ("Tacotron", "Synthesizer"):
        from . import tacotron
        return getattr(tacotron, name)
    if name in ("WaveNetTrainer",):
        from .train import WaveNetTrainer
        return WaveNetTrainer
    if name in ("text_to_wave",):
        from .e2e import text_to_wave
        return text_to_wave
    if name in ("mu_law_encode", "mu_law_decode", "mu_law_expand"):
        from . import ops
        return getattr(ops, name)
    raise AttributeError(name)
