"""ap-adapter_amd: MI355X-native (gfx950) implementation of the AP-adapter audio-conditioned diffusion hot path.

Import as ``ap_adapter_amd`` (the repo-root shim ``ap_adapter_amd.py`` maps that name onto this directory)."""
from .processors import AttnProcessor2_0, IPAttnProcessor2_0  # noqa: F401
from .unet import AudioLDM2UNet2DConditionModel, UNetConfig  # noqa: F401
from .audiomae import AudioMAEConditionCTPoolRand, Vanilla_AudioMAE, AudioMAEEncoder  # noqa: F401
from .scheduler import DDIMScheduler  # noqa: F401
from .pipeline import AudioLDM2Pipeline  # noqa: F401
from .vocoder import SpeechT5HifiGan, HifiGanConfig  # noqa: F401
from .vae import AutoencoderKL, VaeConfig  # noqa: F401
from .text_encoders import (PromptEncoder, ClapTextModelWithProjection, T5EncoderModel, GPT2Model,  # noqa: F401
                            AudioLDM2ProjectionModel)
from .wiring import install_ap_adapter, build_processors, ip_layer_names, adapter_state_dict, save_adapter, load_adapter  # noqa: F401
from . import ops, distributed, autograd, config, sharded  # noqa: F401
from .config import get_config  # noqa: F401
from .training import AdapterTrainer, add_noise  # noqa: F401
from ._lib import build, lib  # noqa: F401
