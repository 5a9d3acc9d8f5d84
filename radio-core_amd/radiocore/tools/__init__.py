"""Imports all modules from radiocore.tools."""

from radiocore.tools.tuner import *
from radiocore.tools.sharding import *
