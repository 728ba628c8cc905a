// Mock of the two DBoW2 container types that cross the front-end boundary (the vendored DBoW2 of the reference defines
// them as std::map subclasses: include/DBoW2/FeatureVector.h:21-22, BowVector.h).  TEST INFRASTRUCTURE.
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class BowVector : public std::map<WordId, WordValue> {};
}  // namespace DBoW2
