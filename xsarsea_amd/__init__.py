"""xsarsea_amd: MI355X-native wind-inversion hot path of xsarsea."""
__version__ = "0.1.0"
