"""Style augmentation surface (reference src/styleaug/styleAugmentor.py:12-68, ghiasi.py:6-136).
The Ghiasi decoder forward (implicit-GEMM 3x3/9x9 convolutions with reflection padding, conditional instance norm) is
the next MFMA-bound row of the hot-path table and is not built yet; the class fails loudly instead of running PyTorch."""


class StyleAugmentor:
    def __init__(self, alpha, device):
        raise NotImplementedError("--randomize_texture (Ghiasi style decoder) has no HIP path yet in this build; "
                                  "see DESIGN.md (scope / next rows)")
