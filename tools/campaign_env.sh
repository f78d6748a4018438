set -e -o pipefail
mkdir -p gpurun_out
RT_LEAF=1 timeout -k 10 200 python -u tools/parity_campaign.py 1500 620000 > gpurun_out/camp_leaf1.log 2>&1
RT_LEAF=4 timeout -k 10 200 python -u tools/parity_campaign.py 1500 622000 > gpurun_out/camp_leaf4.log 2>&1
RT_MESH_IN_LDS=1 timeout -k 10 200 python -u tools/parity_campaign.py 1500 150000 > gpurun_out/camp_staged.log 2>&1
RT_MESH_IN_LDS=1 RT_LEAF=1 RT_GROUP_WAVES=2 timeout -k 10 200 python -u tools/parity_campaign.py 1000 152000 > gpurun_out/camp_staged_leaf1.log 2>&1
RT_GROUP_WAVES=8 timeout -k 10 300 python -u tools/parity_campaign.py --calls 150 612000 > gpurun_out/camp_calls_g8.log 2>&1
