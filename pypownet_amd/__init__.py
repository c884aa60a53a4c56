"""pypownet_amd: MI355X-native load-flow step engine behind pypownet's RunEnv API."""
from .case import ARTIFICIAL_NODE_STARTING_STRING  # noqa: F401

__version__ = '0.1.0'
