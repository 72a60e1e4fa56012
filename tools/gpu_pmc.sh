#!/bin/bash
bash tools/pmc_traffic.sh
