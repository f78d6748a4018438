# Parity campaign under the library's switches, with the camera rays' lists in use at every sample count; first with
# trees that get levels until their top one is a single node (RT_TREE_TOP=1), so that the descent through upper tree
# levels runs on meshes of a few hundred triangles.   bash tools/campaign_r3_env.sh
set -o pipefail
mkdir -p gpurun_out
export RT_LISTS_MIN_SPP=1
export RT_TREE_TOP=1
RT_LEAF=1 timeout -k 10 200 python -u tools/parity_campaign.py 1500 330000 > gpurun_out/r3e_deep_leaf1.log 2>&1; echo "deep trees, leaves of one: rc=$?"; tail -n 1 gpurun_out/r3e_deep_leaf1.log
RT_LEAF=2 timeout -k 10 200 python -u tools/parity_campaign.py 1000 332000 > gpurun_out/r3e_deep_leaf2.log 2>&1; echo "deep trees, leaves of two: rc=$?"; tail -n 1 gpurun_out/r3e_deep_leaf2.log
RT_LEAF=1 timeout -k 10 200 python -u tools/parity_campaign.py --calls 150 670000 > gpurun_out/r3e_deep_calls.log 2>&1; echo "deep trees, calls: rc=$?"; tail -n 1 gpurun_out/r3e_deep_calls.log
unset RT_TREE_TOP
for g in 1 2 3 8; do
  RT_GROUP_WAVES=$g timeout -k 10 150 python -u tools/parity_campaign.py 800 $((340000 + 1000 * g)) > gpurun_out/r3e_group$g.log 2>&1; echo "workgroups of $g waves: rc=$?"; tail -n 1 gpurun_out/r3e_group$g.log
done
RT_MESH_IN_LDS=1 timeout -k 10 200 python -u tools/parity_campaign.py 1500 350000 > gpurun_out/r3e_staged.log 2>&1; echo "mesh staged in LDS: rc=$?"; tail -n 1 gpurun_out/r3e_staged.log
RT_GROUP_WAVES=3 timeout -k 10 250 python -u tools/parity_campaign.py --calls 150 672000 > gpurun_out/r3e_group3_calls.log 2>&1; echo "workgroups of three waves, calls: rc=$?"; tail -n 1 gpurun_out/r3e_group3_calls.log
RT_NO_TILE_LISTS=1 timeout -k 10 150 python -u tools/parity_campaign.py 800 352000 > gpurun_out/r3e_nolists.log 2>&1; echo "without lists: rc=$?"; tail -n 1 gpurun_out/r3e_nolists.log
