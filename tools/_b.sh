#!/bin/bash
python -m gpzoo_amd.build > /tmp/w/build.log 2>&1; rc=$?; grep -E "error|undefined|RuntimeError" /tmp/w/build.log | head -20; echo "build rc=$rc"; ls -la /root/repo/gpzoo_amd/libgpzoo_hip.so
