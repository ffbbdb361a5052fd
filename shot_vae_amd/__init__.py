"""shot_vae_amd: MI355X-native (gfx950 HIP) implementation of the SHOT-VAE training hot path behind
the reference's Python API (FengHZ/SHOT-VAE: shot_vae_model/vae.py, lib/criterion.py,
lib/utils/mixup.py, the step of main_shot_vae.py)."""
from .vae import VariationalAutoEncoder          # noqa: F401
from .criterion import VAECriterion, ClsCriterion, continuous_posterior_loss   # noqa: F401
from .mixup import mixup_vae_data, label_smoothing, optimal_match_index        # noqa: F401
from .optim import FlatSGD, FlatAdam              # noqa: F401
from .train import (train_step, train_step_overlapped, train_step_grouped, GraphedTrainStep, DeviceRng, schedule,   # noqa: F401
                    alpha_schedule, m2_train_step, apply_update, inference_kl)
from .evaluate import Evaluator, evaluate       # noqa: F401
from .data import DeviceDataset, ssl_split      # noqa: F401
from .smooth import (SmoothVAE, svhn_VAE, mnist_VAE, SmoothELBOLoss, smooth_train_step,      # noqa: F401
                     GraphedSmoothStep)
from .trace import StepLogger, step_range, enable_ranges     # noqa: F401
