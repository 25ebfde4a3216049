#!/bin/bash
cd /root/repo && python -m gpzoo_amd.build 2>&1 | grep -E "rror|warning: v" ; ls -la /root/repo/gpzoo_amd/libgpzoo_hip.so
