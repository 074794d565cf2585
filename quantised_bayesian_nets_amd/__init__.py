"""quantised_bayesian_nets_amd -- MI355X-native Monte-Carlo inference path for quantised Bayesian networks.

Drop-in for the hot path of martinferianc/quantised-bayesian-nets (reference file:line citations in each module).
"""
from .layers import Conv2d, ConvReLU2d, Linear, LinearReLU, MCQTensor, QFunctional, mc_context  # noqa: F401
from .models import BasicBlock, ConvNetwork_ResNet, ModelFactory  # noqa: F401
from .models_mc import BernoulliDropout  # noqa: F401
from .mc import mc_predict, mc_predict_regression, shard_samples, finalize_moments, reduce_moments, GraphedPredictor  # noqa: F401
from .quant import UINT_BOUNDS, INT_BOUNDS, NOISE_SCALE, NOISE_ZERO_POINT  # noqa: F401
from .metrics import ClassificationMetric, RegressionMetric  # noqa: F401
from . import models_f32, models_qat, models_mc, models_mc_f32, models_small  # noqa: F401  (float, QAT-eval, MC-Dropout int8 / float, small int8 graphs)
