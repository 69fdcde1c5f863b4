#!/usr/bin/env python3
"""Timing ablations of the wave-specialised three-plane kernel (csrc/conv_ws_x3.hip), one layer shape, every tile:
dbg 0 product | 4 no epilogue | 16384 consumers alone | 32768 barriers but no loads | 65536 loads but no barriers |
16388 consumers alone without epilogue.   python tools/ws_x3_ablate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ws_x3_probe as P  # noqa: E402

D = (0, 4, 16384, 32768, 65536, 16388)
P.probe("Mixed_6e 1x7 192", 384, 17, 17, 192, 192, 1, 7, D)
P.probe("Mixed_5 3x3 96->96", 384, 35, 35, 96, 96, 3, 3, D)
