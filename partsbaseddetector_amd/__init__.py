"""partsbaseddetector_amd -- MI355X-native detection hot path of PartsBasedDetector.

HOG feature pyramid -> filter-bank correlation -> distance-transform dynamic program -> candidates,
as hand-written HIP kernels for gfx950 behind the C ABI in include/pbd.h, with a host-side mirror of
the reference's IFeatures / IConvolutionEngine / DynamicProgram / PartsBasedDetector interface.
"""
from .model import Model, FlatModel, synthetic_model, synthetic_person_model, synthetic_face_model, synthetic_tiny_model  # noqa: F401
from .synth import synthetic_frame  # noqa: F401


def __getattr__(name):
    # the detector classes need the HIP library; import them lazily so that model/synth utilities
    # stay importable, and fail loudly (ImportError) when the library is absent
    if name in ("PartsBasedDetector", "HOGFeatures", "SpatialConvolutionEngine", "DynamicProgram", "Candidate",
                "Handle", "PbdError", "DetectorPool"):
        from . import detector
        return getattr(detector, name)
    raise AttributeError(name)
