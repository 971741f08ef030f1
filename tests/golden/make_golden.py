"""Writes tests/golden/cornell256_B.npz from the CPU oracle (oracle-generated fixture; the reference ships none
and cannot be run here).  Usage: python tests/golden/make_golden.py"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
import ugrt  # noqa: E402

s = ugrt.scenes.cornell()
setup = ugrt.FrameSetup(s["cameras"]["B"], s["light_camera"], s["shading_light"])
r = O.frame(s, setup, 256, 256, light_grid=(128, 128))
pr = r["primary"]
np.savez_compressed(os.path.join(HERE, "cornell256_B.npz"), id=pr["id"].astype(np.int8), t_bits=pr["t"].view(np.uint32),
                    shadowed=r["is_shadowed"].astype(np.uint8), image=r["image"],
                    crc_dir=np.uint32(zlib.crc32(pr["dir"].tobytes())), crc_normal=np.uint32(zlib.crc32(pr["normal"].tobytes())),
                    crc_map=np.uint32(zlib.crc32(r["map"].tobytes())), nchunks=np.int32(r["nchunks"]),
                    R=np.int32(r["grid"]["R"]), lR=np.int32(r["lgrid"]["R"]))
print("wrote cornell256_B.npz", os.path.getsize(os.path.join(HERE, "cornell256_B.npz")), "bytes")
