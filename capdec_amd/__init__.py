"""capdec_amd -- MI355X-native caption hot path of CapDec (CLIP embedding -> noise -> mapping
network -> GPT-2 KV-cached greedy / beam decode) behind the reference's own names.

    from capdec_amd.gpt2_prefix import ClipCaptionModel, MappingType
    from capdec_amd.predictions_runner import generate_beam, generate2

Everything computes in libcapdec_hip.so (hand-written HIP for gfx950) through the C ABI of
include/capdec.h; importing the package does not need a GPU, using it does."""
__version__ = "0.1.0"
