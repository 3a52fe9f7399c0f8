#!/bin/bash
timeout 900 python scripts/r03/par_debug.py 50 2>&1 | tail -20
