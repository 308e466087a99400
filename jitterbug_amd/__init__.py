"""jitterbug_amd - MI355X-native, lockstep-vectorised Jitterbug environment."""
__version__ = "0.1.0"
