import sys, ctypes, os
mode = sys.argv[1]
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
if mode == 'libonly':
    L = g.load_package().lib(); print(mode, 'count', L.mi_blur_device_count(), maps())
elif mode == 'torchfirst':
    import torch; print(torch.cuda.is_available())
    L = g.load_package().lib(); print(mode, 'count', L.mi_blur_device_count(), maps())
elif mode == 'libfirst':
    L = g.load_package().lib()
    import torch; print(torch.cuda.is_available())
    print(mode, 'count', L.mi_blur_device_count(), maps())
elif mode == 'libfirst_count_first':
    L = g.load_package().lib(); print('count before torch', L.mi_blur_device_count())
    import torch; print(torch.cuda.is_available(), torch.cuda.device_count())
    print(mode, 'count', L.mi_blur_device_count(), maps())
