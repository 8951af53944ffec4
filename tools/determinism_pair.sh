#!/bin/bash
# two identical probe processes at the same time on one GPU; prints both verdicts
(timeout 300 python tools/determinism_probe.py ${REPS:-24} > /tmp/pp1.log 2>&1 &)
timeout 300 python tools/determinism_probe.py ${REPS:-24} 2>&1 | tail -1 | cut -c1-160
sleep 4; tail -1 /tmp/pp1.log | cut -c1-160
