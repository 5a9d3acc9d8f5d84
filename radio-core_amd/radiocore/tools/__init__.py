"""Imports all modules from radiocore.tools."""

from radiocore.tools.tuner import *
from radiocore.tools.buffer import *
from radiocore.tools.chopper import *
from radiocore.tools.carrousel import *
from radiocore.tools.ringbuffer import *
from radiocore.tools.sharding import *
from radiocore.tools.wire import *
from radiocore.tools.feeder import *
from radiocore.tools.lanes import *
from radiocore.tools.arena import *
