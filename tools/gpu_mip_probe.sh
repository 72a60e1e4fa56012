#!/bin/bash
mkdir -p gpurun_out/mip && cd gpurun_out/mip
python -c "
import sys; sys.path.insert(0,'../../tests/golden')
import gen_xml_fixtures as g; print(g.write_all('.'))" > gen.log 2>&1
timeout -k 5 200 ../../etol_amd/lib/etol_mi355x_example1 mip_2d_ex1.xml > mip_example.log 2>&1
echo "rc=$?"; grep -v "^ *[0-9]" mip_example.log | tail -40
