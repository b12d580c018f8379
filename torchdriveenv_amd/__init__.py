"""MI355X-native batched driving environment (hot path of inverted-ai/torchdriveenv)."""
__version__ = "0.1.0"
