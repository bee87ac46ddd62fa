from .siren import Siren

# reference: implicit_image/models/__init__.py:5 (fourier / wavelet_siren are outside the hot path)
registry = {"siren": Siren}
