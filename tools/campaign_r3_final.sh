# The parity campaign once more on the final kernel, on seed ranges of its own.   bash tools/campaign_r3_final.sh
set -o pipefail
mkdir -p gpurun_out
RT_LISTS_MIN_SPP=1 timeout -k 10 300 python -u tools/parity_campaign.py 3000 310000 > gpurun_out/r3f_camp_lists.log 2>&1; echo "lists: rc=$?"; tail -n 2 gpurun_out/r3f_camp_lists.log
RT_LISTS_MIN_SPP=1 timeout -k 10 200 python -u tools/parity_campaign.py 1500 40000 > gpurun_out/r3f_camp_lists_plain.log 2>&1; echo "lists, plain seeds: rc=$?"; tail -n 2 gpurun_out/r3f_camp_lists_plain.log
timeout -k 10 120 python -u tools/parity_campaign.py 1000 720000 > gpurun_out/r3f_camp_crowded.log 2>&1; echo "crowded: rc=$?"; tail -n 2 gpurun_out/r3f_camp_crowded.log
timeout -k 10 300 python -u tools/parity_campaign.py --calls 500 650000 > gpurun_out/r3f_camp_calls.log 2>&1; echo "calls: rc=$?"; tail -n 2 gpurun_out/r3f_camp_calls.log
timeout -k 10 300 python -u tools/parity_campaign.py --calls --big 60 660000 > gpurun_out/r3f_camp_calls_big.log 2>&1; echo "calls big: rc=$?"; tail -n 2 gpurun_out/r3f_camp_calls_big.log
