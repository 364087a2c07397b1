"""ideas_amd — MI355X-native hot path of IDEAS."""
__version__ = "0.1.0"
