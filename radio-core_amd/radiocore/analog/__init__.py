"""Imports all modules from radiocore.analog."""

from radiocore.analog.pll import *
from radiocore.analog.wbfm import *
from radiocore.analog.mfm import *
from radiocore.analog.fm import *
from radiocore.analog.deemphasis import *
from radiocore.analog.decimate import *
from radiocore.analog.bandpass import *
