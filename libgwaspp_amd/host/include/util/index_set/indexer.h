// util/index_set/indexer.h -- the part of the reference's row/column index
// interface (src/util/index_set/indexer.h:38-71) that the table path uses:
// tables read included_size() in their constructor (util/table/table.h:51) and
// name rows/columns through getIDAtOrderedIndex().  The include/exclude
// machinery of the reference's IndexedSet is bookkeeping outside the scan path
// and is not mirrored.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

namespace util {

class indexer {
public:
    virtual int orderOf(const std::string &id) = 0;
    virtual std::string getIDAtOrderedIndex(int ord) = 0;
    virtual int included_size() const = 0;
    virtual int maximum_size() const = 0;
    virtual ~indexer() {}
};

// n entries in file order; ids default to their ordinal
class SequentialIndexer : public indexer {
public:
    explicit SequentialIndexer(int n) : n_(n) {}
    explicit SequentialIndexer(const std::vector<std::string> &ids) : n_((int)ids.size()), ids_(ids) {}
    int orderOf(const std::string &id) {
        if (ids_.empty()) {
            try { int v = std::stoi(id); return (v >= 0 && v < n_) ? v : -1; } catch (...) { return -1; }
        }
        if (order_.empty())
            for (int i = 0; i < n_; ++i) order_.emplace(ids_[i], i);  // first occurrence wins
        auto it = order_.find(id);
        return it == order_.end() ? -1 : it->second;
    }
    std::string getIDAtOrderedIndex(int ord) { return ids_.empty() ? std::to_string(ord) : ids_[ord]; }
    int included_size() const { return n_; }
    int maximum_size() const { return n_; }

private:
    int n_;
    std::vector<std::string> ids_;
    std::unordered_map<std::string, int> order_;
};

}  // namespace util
