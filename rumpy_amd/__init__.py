"""rumpy_amd - MI355X-native (gfx950) drop-in for the convolutional SR hot path of um-dsrg/RUMpy:
EDSR / RCAN train + eval step behind the rumpy/SISR model-handler plugin API.  See DESIGN.md."""
__version__ = '0.1.0'
