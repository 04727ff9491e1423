#!/usr/bin/env python3
"""unmicst-solo entry point: same command line as the reference's UnMicst1-5.py:713-876, served by the HIP engine
(unmicst_amd.driver / unmicst_amd.unet2d.UNet2D -> libumx)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.realpath(__file__)))

if __name__ == "__main__":
    from unmicst_amd import driver
    driver.main("unmicst-solo", __file__)
