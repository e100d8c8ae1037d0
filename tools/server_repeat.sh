#!/bin/bash
# the end-to-end server test a few times over (a visibility race between the drain's deliveries and the header shows as a
# wrong receive message now and then)
for i in 1 2 3 4 5 6; do timeout -k 10 300 python -m pytest tests/test_gpu_server.py tests/test_gpu_events.py -x -q 2>&1 | tail -n 1; done
