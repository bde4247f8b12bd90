// Utils.cpp -- see Utils.h.  Host-only; no device code involved.
#include "Utils.h"

namespace cuHE_Utils {

// fields of `data` separated by any run of characters of `seps` (strtok semantics, cuhe/Utils.cu:79-91)
static vector<string> splitFields(const string &data, const string &seps) {
	vector<string> out;
	size_t pos = 0;
	while (pos < data.size()) {
		const size_t a = data.find_first_not_of(seps, pos);
		if (a == string::npos) break;
		size_t b = data.find_first_of(seps, a);
		if (b == string::npos) b = data.size();
		out.push_back(data.substr(a, b - a));
		pos = b;
	}
	return out;
}

Picklable::Picklable(string k, ZZ *cs, int l) : key(k), coeffs(NULL), coeffs_len(0) {
	for (int i = l - 1; i >= 0; i--) SetCoeff(poly, i, cs[i]);
	toCoeffs();
	setValuesString();
}
Picklable::Picklable(string k, ZZX p) : key(k), poly(p), coeffs(NULL), coeffs_len(0) {
	toCoeffs();
	setValuesString();
}
Picklable::Picklable(string data) : coeffs(NULL), coeffs_len(0) { fromString(data); }
Picklable::Picklable(string data, string sep) : coeffs(NULL), coeffs_len(0), separator(sep) { fromString(data); }
Picklable::Picklable(const Picklable &o) : key(o.key), poly(o.poly), coeffs(NULL), coeffs_len(0), separator(o.separator) {
	toCoeffs();
	setValuesString();
}
Picklable::~Picklable() { delete[] coeffs; }

void Picklable::toCoeffs() {
	delete[] coeffs;
	coeffs_len = (int)deg(poly) + 1;
	coeffs = new ZZ[coeffs_len > 0 ? coeffs_len : 1];
	for (int i = 0; i < coeffs_len; i++) coeffs[i] = coeff(poly, i);
}
void Picklable::fromString(const string &data) {
	const vector<string> f = splitFields(data, separator);
	clear(poly);
	if (!f.empty()) key = f[0];
	for (size_t i = f.size(); i-- > 1;) SetCoeff(poly, (long)i - 1, conv<ZZ>(f[i].c_str()));
	toCoeffs();
	setValuesString();
}
void Picklable::setValuesString() {
	stringstream buffer;
	for (int i = 0; i < coeffs_len; i++) {
		buffer << coeffs[i];
		if (i != coeffs_len - 1) buffer << separator;
	}
	values = buffer.str();
}

PicklableMap::PicklableMap(vector<Picklable *> ps) : picklables(ps) {}
PicklableMap::PicklableMap(string data) { fromString(data, ","); }
PicklableMap::PicklableMap(string data, string psep) { fromString(data, psep); }
PicklableMap::PicklableMap(string data, string sep, string psep) : separator(sep) { fromString(data, psep); }
// entries are not deleted: callers keep using what get() handed out after the map is gone (DHS.cu:62-70), as with
// the reference (Utils.cu:186-188)
PicklableMap::~PicklableMap() { picklables.clear(); }

void PicklableMap::fromString(const string &data, const string &psep) {
	picklables.clear();
	for (const string &entry : splitFields(data, separator)) picklables.push_back(new Picklable(entry, psep));
}
string PicklableMap::toString() {
	string out;
	for (size_t i = 0; i < picklables.size(); i++) {
		out += picklables[i]->pickle();
		if (i != picklables.size() - 1) out += separator;
	}
	return out;
}
Picklable *PicklableMap::get(string key) {
	for (size_t i = 0; i < picklables.size(); i++)
		if (picklables[i]->getKey() == key) return picklables[i];
	throw "not found";
}

} // namespace cuHE_Utils
