"""fsnet_amd — MI355X-native (gfx950) implementation of FSNet's self-supervised monodepth
training step behind FSNet's own cfg/builder plugin surface.  See DESIGN.md."""
__version__ = "0.1.0"
