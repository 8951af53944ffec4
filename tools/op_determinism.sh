#!/bin/bash
(timeout 500 python tools/op_determinism.py ${REPS:-200} > /tmp/od1.log 2>&1 &)
timeout 500 python tools/op_determinism.py ${REPS:-200} 2>&1 | grep differ
echo "--- second process"; sleep 3; grep differ /tmp/od1.log
