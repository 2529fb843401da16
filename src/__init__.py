"""Import aliases with the reference's package layout (src.nets, src.core, src.styleaug, src.utils.utils)."""
