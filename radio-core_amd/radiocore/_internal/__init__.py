"""Imports all modules from radiocore._internal."""

from radiocore._internal.injector import *
