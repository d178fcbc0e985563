"""MI355X-native flow-transform hot path (MADE conditioner -> affine / RQ-spline transform ->
per-sample log|det J|) behind the reference's module API.  See DESIGN.md."""
from . import _lib
from ._lib import invalidate_weight_caches
from .made import MADE, MaskedLinear, DenseLinear, MaskedResidualBlock, MaskedFeedforwardBlock, \
    set_matmul_dtype, get_matmul_dtype
from .resnet import ResidualNet, ResidualBlock
from .transforms import CompositeTransform, AutoregressiveTransform, MaskedAffineAutoregressiveTransform, \
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform, MaskedPiecewiseLinearAutoregressiveTransform, \
    MaskedPiecewiseQuadraticAutoregressiveTransform, MaskedPiecewiseCubicAutoregressiveTransform, \
    PiecewiseCubicCouplingTransform, CouplingTransform, AffineCouplingTransform, \
    PiecewiseRationalQuadraticCouplingTransform, PiecewiseRationalQuadraticCDF, PiecewiseCubicCDF
from . import parallel, metrics
from .checkpoint import save_checkpoint, load_checkpoint
from .flows import MaskedAutoregressiveFlow, SimpleMADE, SimpleMaskedLinear, apply_flow_to_samples, \
    apply_reverse_to_samples
from . import data
from .data import two_moons
from .distributions import Distribution, StandardNormal, Uniform, PlummerModel, BaseNeuralSplineFlow, Flow

__all__ = [n for n in dir() if not n.startswith('_')]
