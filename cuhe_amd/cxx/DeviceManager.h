// DeviceManager.h -- device count + pooled allocator entry points
// (API of cuhe/DeviceManager.h; pool itself lives behind cuhe_hip_malloc/free).
#pragma once
#include <cstddef>

namespace cuHE {

void setNumDevices(int val);
int numDevices();

void bootDeviceAllocator(size_t blockBytes, unsigned long num = 0);
void haltDeviceAllocator();
bool deviceAllocatorIsOn();
// allocate / release on the device the calling thread last selected through this API
void *deviceMalloc(size_t size);
void deviceFree(void *ptr);
// which device deviceMalloc/deviceFree act on (the reference uses cudaGetDevice)
void selectDevice(int dev);
int selectedDevice();

} // namespace cuHE
