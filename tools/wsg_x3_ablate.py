#!/usr/bin/env python3
"""Timing ablations of the GEMM mode (csrc/conv_ws_x3.hip) on two sibling GEMMs: dbg 0 product | 4 no epilogue | 16384
consumers alone | 32768 barriers and A tiles but no filter DMA | 65536 loads but no barriers.   python tools/wsg_x3_ablate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wsg_x3_probe as P  # noqa: E402

D = (0, 4, 16384, 32768, 65536)
P.probe("Mixed_6e siblings", 384, 12, 768, 768, D)
P.probe("Mixed_5d siblings", 384, 25, 288, 240, D)
