#!/usr/bin/env python3
"""Wall time of creating a 1 GiB text on the device, generator + alphabet pass (api.cpp text_alphabet): python tools/alphabet_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, smart_amd
from smart_amd import Text
for sigma in (2, 4, 128, 256):
    for rep in range(2):
        t0=time.time(); t=Text.generate(0x5EED0001, sigma, 1<<30); dt=time.time()-t0
        a=t.alphabet(); t.free()
        print("sigma %3d: generate + alphabet of 1 GiB %.2f ms, %d values"%(sigma, dt*1e3, len(a)))
