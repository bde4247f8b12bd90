// DeviceManager.h -- device count + pooled allocator entry points
// (API of cuhe/DeviceManager.h; pool itself lives behind cuhe_hip_malloc/free).
#pragma once
#include <cstddef>

#include <map>

namespace cuHE {

// The reference's per-device block allocator type (cuhe/DeviceManager.h:36-52).  The pool itself lives behind the C ABI
// (cuhe_hip_malloc / cuhe_hip_free: size-keyed, shared by everything on the device); an object of this class is a view of
// it for the device that was selected when it was made, and remembers what it handed out so that freeAll() / the
// destructor can give it back.
class DeviceAllocator {
public:
	DeviceAllocator();
	~DeviceAllocator();
	char *allocate(std::ptrdiff_t size);
	void deallocate(char *ptr);
	void freeAll();
private:
	int device_;
	std::map<char *, std::ptrdiff_t> allocatedBlocks;
};

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
