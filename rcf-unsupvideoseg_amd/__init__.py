"""MI355X-native implementation of RCF's per-frame training hot path."""
__version__ = "0.1.0"
