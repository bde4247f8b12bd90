// Parameters.h -- the global `cuHE::param` the scheme code reads directly
// (examples/DHS/DHS.cu:36-54 in the reference).  API of cuhe/Parameters.h:34-78;
// every value is derived inside the C-ABI library (cuhe_hip_set_parameters) and
// mirrored here, so the host and the device side can never disagree.
#pragma once

namespace cuHE {

struct GlobalParameters {
	// ring
	int mSize, modLen, modLen2, rawLen, crtLen, nttLen;
	// coefficient-modulus chain
	int logCoeffMax, logCoeffMin, logCoeffCut;
	// circuit
	int depth, modMsg, logMsg, wordsMsg;
	// relinearisation
	int logRelin, numEvalKey;
	// CRT
	int logCrtPrime, numCrtPrime;
	// level-dependent values (lvl == -1: plaintext)
	int _numCrtPrime(int lvl);
	int _logCoeff(int lvl);
	int _wordsCoeff(int lvl);
	int _numEvalKey(int lvl);
	int _getLevel(int logq);
};

extern GlobalParameters param;

void setParam(int d, int p, int w, int min, int cut, int m);
void resetParam();

} // namespace cuHE
