#!/usr/bin/env python
"""bench.py's driver legs (sso over the fixture x 100, classic over 8 BAMs) at several block sizes of the bulk route
(SVT_BULK_BLOCK_SITES): how much of the reader the other stages hide.  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for units in (int(x) for x in (sys.argv[1:] or ["4096", "16384", "65536", "400000"])):
    os.environ["SVT_BULK_BLOCK_SITES"] = str(units)
    legs = bench.driver_legs()
    print("block %7d units | sso %.1f ms = %.0f sites/s | classic 8 BAMs %.1f ms = %.0f units/s | same bytes %s %s" % (
        units, legs["driver_sso"]["wall_ms"], legs["driver_sso"]["sites_per_s"], legs["driver_classic_8bam"]["wall_ms"],
        legs["driver_classic_8bam"]["units_per_s"], legs["driver_sso"]["per_line_route"]["same_bytes"],
        legs["driver_classic_8bam"]["per_line_route"]["same_bytes"]), flush=True)
