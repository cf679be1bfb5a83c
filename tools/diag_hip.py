import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l or 'libjsg' in l})
if mode == 'plain':
    l = ctypes.CDLL(os.path.join(os.path.dirname(__file__), '..', 'jadespectrogram_amd', 'libjsg.so'))
    print('plain device_count', l.jsg_device_count()); print(maps())
elif mode == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
    l = ctypes.CDLL(os.path.join(os.path.dirname(__file__), '..', 'jadespectrogram_amd', 'libjsg.so'))
    print('after torch device_count', l.jsg_device_count()); print(maps())
elif mode == 'torch_import_only':
    import torch
    l = ctypes.CDLL(os.path.join(os.path.dirname(__file__), '..', 'jadespectrogram_amd', 'libjsg.so'))
    print('torch imported, device_count', l.jsg_device_count()); print(maps())
