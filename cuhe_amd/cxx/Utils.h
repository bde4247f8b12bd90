// Utils.h -- text serialisation of ZZ arrays and ZZX polynomials: the key-file format of the reference's
// cuhe/Utils.h:38-93 (used by examples/DHS/DHS.cu:62-124 to save and reload keys).
//
// Wire format (cuhe/Utils.cu:76-121,141-146,203-213), kept byte for byte:
//   one entry  = key SEP c0 SEP c1 SEP ... SEP c_deg        decimal integers, low degree first, SEP = ","
//   a map      = entries joined by "\n"                      (no trailing separator)
// Parsing splits on any run of separator characters (strtok semantics: empty fields vanish), the first field is
// the key.  A polynomial drops its trailing zero coefficients; an array of ZZ goes through a polynomial too
// (Utils.cu:34-37), so trailing zeros of an array vanish as well.
//
// Ownership differs from the reference where the reference is undefined behaviour (its destructor runs
// `delete &key`, Utils.cu:113-114, and frees the CALLER's array): an entry owns copies of what it was given, and
// getCoeffs() returns an array that stays valid for the life of the entry.
#pragma once
#include <NTL/ZZ.h>
#include <NTL/ZZX.h>
#include <sstream>
#include <string>
#include <vector>
NTL_CLIENT

namespace cuHE_Utils {

class Picklable {
	string key;
	string values;
	ZZX poly;
	ZZ *coeffs;
	int coeffs_len;
	string separator = ",";

public:
	Picklable(string key, ZZ *coeffs, int len);
	Picklable(string key, ZZX poly);
	Picklable(string data);                    // parse "key,c0,c1,..."
	Picklable(string data, string sep);
	Picklable(const Picklable &);
	~Picklable();

	void setSeparator(string s) { separator = s; setValuesString(); }
	string getSeparator() { return separator; }

	ZZX getPoly() { return poly; }
	ZZ *getCoeffs() { return coeffs; }             // owned by this entry
	int getCoeffsLen() { return coeffs_len; }

	string getKey() { return key; }
	string getValues() { return values; }

	string pickle() { return key + separator + values; }      // "key,c0,c1,..."

private:
	Picklable &operator=(const Picklable &);
	void setValuesString();
	void toCoeffs();
	void fromString(const string &);
};

class PicklableMap {
	vector<Picklable *> picklables;
	string separator = "\n";

public:
	PicklableMap(vector<Picklable *>);
	PicklableMap(string data);
	PicklableMap(string data, string psep);
	PicklableMap(string data, string sep, string psep);
	~PicklableMap();

	void setSeparator(string sep) { separator = sep; }
	string getSeparator() { return separator; }

	vector<Picklable *> getPicklables() { return picklables; }
	string toString();

	Picklable *get(string key);                // throws (const char *)"not found" (DHS.cu:85-90 relies on it)

private:
	void fromString(const string &, const string &psep);
};

} // namespace cuHE_Utils
