#!/bin/bash
# Probe of the GPU box for anything that could pin the oracle (SURVEY §7 hard part 1):
# OpenCV (python cv2 or C++ headers/libs), Eigen, Sophus, g2o, DBoW2, plus host facts.
out=gpurun_out/probe_box.txt
{
echo "== date"; date
echo "== nproc / cpu"; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
echo "== python cv2"; python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
echo "== python skimage/PIL/scipy.ndimage"; python -c "import skimage; print('skimage', skimage.__version__)" 2>&1 | tail -1; python -c "import PIL; print('PIL', PIL.__version__)" 2>&1 | tail -1
echo "== opencv headers"; find / -xdev \( -name "opencv2" -o -name "opencv*.pc" -o -name "libopencv_core*" \) 2>/dev/null | head
echo "== eigen"; find / -xdev \( -path "*/Eigen/Dense" -o -name "eigen3*.pc" -o -name "signature_of_eigen3_matrix_library" \) 2>/dev/null | head
echo "== sophus / g2o / DBoW2"; find / -xdev \( -iname "sophus" -o -iname "g2o" -o -iname "DBoW2" \) 2>/dev/null | head
echo "== pkg-config"; pkg-config --list-all 2>/dev/null | grep -i -E "opencv|eigen" | head
echo "== rccl"; ls /opt/rocm/lib/librccl* 2>/dev/null
echo "== gpus"; rocm-smi --showid 2>/dev/null | head -20
} > $out 2>&1
cat $out
