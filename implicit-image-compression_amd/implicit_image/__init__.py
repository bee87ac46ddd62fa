"""MI355X-native host mirror of the reference's `implicit_image` hot-path interface.

Same module and function names as the reference for the path this build replaces
(`data.get_grid`, `models.registry["siren"]`, `utils.train_helper.train_epoch / eval_epoch /
get_optimizer_lr_scheduler / setup_mask`), backed by the gfx950 engine in `csrc/libsiren_fit.so`
through the C ABI declared in `include/siren_fit.h`.  There is no CPU fallback: importing
`implicit_image._engine` without the built library, or creating an engine without a gfx950 device,
raises.
"""
__all__ = ["_engine"]
