# A parity campaign inside one GPU call: random scenes with the camera rays' candidate lists in use at every
# sample count (RT_LISTS_MIN_SPP=1), then every way of calling the renderer.   bash tools/campaign_r3.sh
set -o pipefail
mkdir -p gpurun_out
RT_LISTS_MIN_SPP=1 timeout -k 10 400 python -u tools/parity_campaign.py 3000 300000 > gpurun_out/r3_camp_lists.log 2>&1; echo "lists: rc=$?"; tail -2 gpurun_out/r3_camp_lists.log
RT_LISTS_MIN_SPP=1 timeout -k 10 300 python -u tools/parity_campaign.py 1500 30000 > gpurun_out/r3_camp_lists_plain.log 2>&1; echo "lists, plain seeds: rc=$?"; tail -2 gpurun_out/r3_camp_lists_plain.log
timeout -k 10 250 python -u tools/parity_campaign.py 1000 710000 > gpurun_out/r3_camp_crowded.log 2>&1; echo "crowded: rc=$?"; tail -2 gpurun_out/r3_camp_crowded.log
timeout -k 10 300 python -u tools/parity_campaign.py --calls 500 630000 > gpurun_out/r3_camp_calls.log 2>&1; echo "calls: rc=$?"; tail -2 gpurun_out/r3_camp_calls.log
timeout -k 10 300 python -u tools/parity_campaign.py --calls --big 120 640000 > gpurun_out/r3_camp_calls_big.log 2>&1; echo "calls big: rc=$?"; tail -2 gpurun_out/r3_camp_calls_big.log
